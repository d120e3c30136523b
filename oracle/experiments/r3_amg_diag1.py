import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle.proto_amg import problem, block_diag_inv, lam_max
from oracle.proto_sa import setup, cycle, run, describe
cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
print("dofs", A.shape[0])
# two-grid, exact coarse solve
for sa in [(), (0,)]:
    lv = setup(A, free, sa_levels=sa, max_levels=2)
    print("two-grid", "SA" if sa else "plain", describe(lv))
    for sch in [(1,), (2,), (3,), (4,)]:
        run(A, b, lv, f"  2-grid nu={sch[0]}", sch, sch)
# smoother only (no coarse), to see what Krylov+Jacobi does
