Optimal - objective value 0.00000000
      6 x6 1 0
     10 x10 1 0
     12 x12 1 0
     34 x34 1 0
