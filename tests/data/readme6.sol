Optimal - objective value 0.00000000
     26 x26                     1                       0
     29 x29                     1                       0
     31 x31                     1                       0
     33 x33                     1                       0
