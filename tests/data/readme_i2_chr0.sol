Optimal - objective value 0.00000000
      3 x3 1 0
     16 x16 1 0
      8 x8 1 0
