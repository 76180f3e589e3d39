Optimal - objective value 0.00000000
     13 x13 1 0
     16 x16 1 0
