Optimal - objective value 0.00000000
      4 x4 1 0
      9 x9 1 0
     30 x30 1 0
