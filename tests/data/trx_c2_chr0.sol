Optimal - objective value 0.00000000
      3 x3 1 0
      5 x5 1 0
      6 x6 1 0
