* hand-written instance exercising every row sense, RANGES on G/L/E(+/-), and the bound types
NAME          FEATURES
ROWS
 N  OBJ
 G  G1
 L  L1
 E  E1
 G  GR
 L  LR
 E  EP
 E  EM
 G  NORHS
COLUMNS
    XA        OBJ        1.0   G1     1.0
    XA        L1         2.0   E1     1.0
    XB        OBJ       -2.0   G1     1.0
    XB        GR         1.0   LR     1.0
    XC        OBJ        0.5   EP     1.0
    XC        EM         1.0   L1    -1.0
    XD        E1         1.0   NORHS  1.0
    XD        GR        -1.0
    XE        OBJ        3.0   LR     2.0
    XE        EP        -1.0   EM     0.5
* a comment in the middle
RHS
    RHS       OBJ        7.0   G1     1.0
    RHS       L1         8.0   E1     3.0
    RHS       GR         1.0   LR     6.0
    RHS       EP         2.0   EM     4.0
RANGES
    RNG       GR         2.5   LR     3.0
    RNG       EP         1.5   EM    -2.0
BOUNDS
 LO BND       XA        -1.0
 UP BND       XA         4.0
 UP BND       XB         5.0
 FX BND       XC         2.0
 FR BND       XD
 MI BND       XE
ENDATA
