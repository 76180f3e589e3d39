Optimal - objective value 0.00000000
      8 x8 1 0
     10 x10 1 0
