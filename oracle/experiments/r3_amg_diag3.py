import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle.proto_amg import problem
from oracle.proto_sa import setup, describe
cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
print("dofs", A.shape[0], "Re", Re, flush=True)
lv = setup(A, free, max_levels=2)
L, C = lv
print(describe(lv), "lam", L.lam)
def twogrid(smooth_pre, smooth_post):
    def f(v):
        x = smooth_pre(v)
        r = v - L.A @ x
        x = x + L.P @ C.lu.solve(L.R @ r)
        return smooth_post(x, v)
    return f
def run(label, f):
    its=[0]
    M_ = spla.LinearOperator(A.shape, matvec=f)
    x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=M_, maxiter=300, callback=lambda xk: its.__setitem__(0, its[0]+1))
    print(f"{label:50s} its {its[0]} info {info} rel {np.linalg.norm(b-A@x)/np.linalg.norm(b):.1e}", flush=True)
def jac(om, nu):
    pre = lambda v: sum_sweeps(np.zeros_like(v), v, om, nu)
    post = lambda x, v: sum_sweeps(x, v, om, nu)
    return pre, post
def sum_sweeps(x, v, om, nu):
    for _ in range(nu):
        x = x + om * (L.Dinv @ (v - L.A @ x))
    return x
for om in [0.5, 0.6, 0.7, 0.8, 0.9]:
    run(f"bjacobi om={om} nu=1", twogrid(*jac(om, 1)))
# block GS forward (pre) / backward (post)
Ab = L.A.tobsr((4,4))
n = L.n
Lo = sp.tril(L.A, 0).tocsr()   # point lower incl diag (approx of block GS)
Up = sp.triu(L.A, 0).tocsr()
def gs_f(x, v, nu=1):
    for _ in range(nu):
        x = x + spla.spsolve_triangular(Lo, v - L.A @ x, lower=True)
    return x
def gs_b(x, v, nu=1):
    for _ in range(nu):
        x = x + spla.spsolve_triangular(Up, v - L.A @ x, lower=False)
    return x
run("point GS fwd pre / fwd post", twogrid(lambda v: gs_f(np.zeros_like(v), v), lambda x, v: gs_f(x, v)))
run("point GS fwd pre / bwd post", twogrid(lambda v: gs_f(np.zeros_like(v), v), lambda x, v: gs_b(x, v)))
ilu = spla.spilu(sp.csc_matrix(L.A), fill_factor=1.0, drop_tol=0.0, permc_spec="NATURAL", diag_pivot_thresh=0.0)
def il(x, v):
    return x + ilu.solve(v - L.A @ x)
run("ILU(~0) pre+post", twogrid(lambda v: il(np.zeros_like(v), v), il))
run("no coarse, ILU only", lambda v: ilu.solve(v))
run("no coarse, bjacobi only", lambda v: L.Dinv @ v)
# Chebyshev degree 2,3 on [lam/a, 1.1 lam]
def cheb(x, v, deg, ratio):
    lmax = 1.1 * L.lam; lmin = L.lam / ratio
    d = (lmax + lmin) / 2; c = (lmax - lmin) / 2
    r = L.Dinv @ (v - L.A @ x)
    p = r / d; x = x + p; alpha = 1.0 / d
    for k in range(1, deg):
        r = L.Dinv @ (v - L.A @ x)
        beta = (c * alpha / 2) ** 2 if k > 1 else 0.5 * (c * alpha) ** 2
        alpha = 1.0 / (d - beta / alpha)
        p = alpha * r + beta * alpha / alpha * p if False else alpha * r + beta * p
        x = x + p
    return x
for deg in [2, 3]:
    for ratio in [3, 5, 10]:
        run(f"cheb deg={deg} ratio={ratio}", twogrid(lambda v: cheb(np.zeros_like(v), v, deg, ratio), lambda x, v: cheb(x, v, deg, ratio)))
