import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle.proto_amg import problem
from oracle.proto_sa import setup, cycle, run, describe
cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
print("dofs", A.shape[0], "Re", Re, flush=True)
for name, sa in [("plain", ()), ("SA@0", (0,)), ("SA all", (0,1,2,3,4,5))]:
    t0=time.time()
    lv = setup(A, free, sa_levels=sa)
    print(name, describe(lv), f"setup {time.time()-t0:.0f}s", flush=True)
    for sch in [(1,4,6,2), (1,2,2,2), (1,1,1,1)]:
        run(A, b, lv, f"{name} V {sch}", sch, sch)
