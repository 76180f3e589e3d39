Optimal - objective value 0.00000000
      2 x2 1 0
      4 x4 1 0
     10 x10 1 0
