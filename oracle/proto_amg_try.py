import sys, time, numpy as np
import oracle.proto_amg as PA
from oracle.proto_amg import *
cells = tuple(int(a) for a in sys.argv[1:4])
Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
levels = setup(A, free)
print("cells", cells, "levels", [L.n for L in levels], flush=True)
run(A, b, levels, "V (1,4,6,2)")
run(A, b, levels, "W at 1,2", kind="W")
run(A, b, levels, "W at 1,2,3,4", kind="W", klevels=(1, 2, 3, 4))
two = setup(A, free, max_levels=2)
run(A, b, two, "two-grid exact coarse nu=1", sched=(1,))
three = setup(A, free, max_levels=3)
run(A, b, three, "three-grid exact (1,4)", sched=(1, 4))
