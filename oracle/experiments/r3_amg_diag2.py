import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle.proto_amg import problem
from oracle.proto_sa import setup, cycle, run, describe
cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
print("dofs", A.shape[0], flush=True)
lv = setup(A, free)
print("plain", describe(lv), flush=True)
for sch in [(1,4,6,2), (1,2,2,2), (1,3,3,2), (1,1,1,1)]:
    run(A, b, lv, f"plain V {sch}", sch, sch)
lv = setup(A, free, sa_levels=(1,2,3,4,5,6))
print("SA>=1", describe(lv), flush=True)
for sch in [(1,4,6,2), (1,2,2,2), (1,1,1,1), (1,2,1,1), (1,3,2,2)]:
    run(A, b, lv, f"SA>=1 V {sch}", sch, sch)
lv = setup(A, free, sa_levels=(1,2,3,4,5,6), restrict_smooth=False)
print("SA>=1 P only", describe(lv), flush=True)
for sch in [(1,2,2,2), (1,1,1,1)]:
    run(A, b, lv, f"SA>=1 Ponly V {sch}", sch, sch)
lv = setup(A, free, sa_levels=(1,))
print("SA==1", describe(lv), flush=True)
for sch in [(1,2,2,2), (1,2,4,2), (1,1,1,1)]:
    run(A, b, lv, f"SA==1 V {sch}", sch, sch)
