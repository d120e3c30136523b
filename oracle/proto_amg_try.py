import sys, time, numpy as np
import scipy.sparse.linalg as spla
import oracle.proto_amg as PA
from oracle.proto_amg import *
cells = tuple(int(a) for a in sys.argv[1:4])
Re = 200.0 * cells[1] / 75.0
A, b, free = problem(cells, Re)
levels = setup(A, free)
print("levels", [L.n for L in levels], flush=True)
run(A, b, levels, "V (1,4,6,2) current")
L = levels[0]
def count(label, f):
    its = [0]
    M_ = spla.LinearOperator(A.shape, matvec=f)
    x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=M_, maxiter=300, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    print(f"{label}: its {its[0]} info {info}", flush=True)
def v01(v):                      # coarse correction on b, then one post sweep: ONE fine matrix pass
    x = L.P @ cycle(levels, 1, L.P.T @ v)
    return x + L.omega * (L.Dinv @ (v - L.A @ x))
def v10(v):                      # free pre-smooth, residual, coarse correction, no post sweep: ONE fine pass
    x = L.omega * (L.Dinv @ v)
    return x + L.P @ cycle(levels, 1, L.P.T @ (v - L.A @ x))
def add(v):                      # additive: no fine pass
    return L.omega * (L.Dinv @ v) + L.P @ cycle(levels, 1, L.P.T @ v)
def v02(v):
    x = L.P @ cycle(levels, 1, L.P.T @ v)
    x = x + L.omega * (L.Dinv @ (v - L.A @ x))
    return x + L.omega * (L.Dinv @ (v - L.A @ x))
count("V(0,1)  one fine pass", v01)
count("V(1,0)  one fine pass", v10)
count("additive, no fine pass", add)
count("V(0,2)  two fine passes", v02)
