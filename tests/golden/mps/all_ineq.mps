NAME ALLINEQ
ROWS
 N  COST
 G  C1
 L  C2
COLUMNS
    X1  COST  1.0  C1  1.0
    X1  C2    1.0
    X2  COST  1.0  C1  2.0
    X2  C2    -1.0
RHS
    R   C1  2.0  C2  3.0
ENDATA
