import sys, time, numpy as np
import scipy.sparse.linalg as spla
import oracle.proto_amg as PA
from oracle.proto_amg import *
cells = tuple(int(a) for a in sys.argv[1:4])
Re = 200.0 * cells[1] / 75.0
A, b, free = problem(cells, Re)
levels = setup(A, free)
M = lambda v: cycle(levels, 0, v)
nb = np.linalg.norm(b)

def idrs(A, b, M, s=4, tol=1e-8, maxit=400, seed=0):
    """IDR(s) biortho variant (van Gijzen & Sonneveld, ACM TOMS Alg. 913), right... here left-applied M as in the paper's
    preconditioned form: v = M (r - G c); counts preconditioner+matvec pairs."""
    n = len(b); rng = np.random.default_rng(seed)
    P = np.linalg.qr(rng.normal(size=(n, s)))[0]
    x = np.zeros(n); r = b.copy(); nr = np.linalg.norm(r)
    G = np.zeros((n, s)); U = np.zeros((n, s)); Ms = np.eye(s); om = 1.0
    pairs = 0
    while nr > tol * nb and pairs < maxit:
        f = P.T @ r
        for k in range(s):
            c = np.linalg.solve(Ms[k:, k:], f[k:])
            v = r - G[:, k:] @ c
            v = M(v)
            U[:, k] = U[:, k:] @ c + om * v
            G[:, k] = A @ U[:, k]; pairs += 1
            for i in range(k):
                al = (P[:, i] @ G[:, k]) / Ms[i, i]
                G[:, k] -= al * G[:, i]; U[:, k] -= al * U[:, i]
            Ms[k:, k] = P[:, k:].T @ G[:, k]
            be = f[k] / Ms[k, k]
            r = r - be * G[:, k]; x = x + be * U[:, k]
            nr = np.linalg.norm(r)
            if nr <= tol * nb: return x, pairs
            if k + 1 < s: f[k + 1:] = f[k + 1:] - be * Ms[k + 1:, k]
        v = M(r); t = A @ v; pairs += 1
        om = (t @ r) / (t @ t)
        # "maintaining the convergence" safeguard
        rho = abs(t @ r) / (np.linalg.norm(t) * np.linalg.norm(r))
        if rho < 0.7: om *= 0.7 / rho
        x = x + om * v; r = r - om * t; nr = np.linalg.norm(r)
    return x, pairs

cnt = [0]
def Mc(v):
    cnt[0] += 1
    return M(v)
Mop = spla.LinearOperator(A.shape, matvec=Mc)
x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=Mop, maxiter=300)
print("bicgstab pairs", cnt[0], "rel", np.linalg.norm(b - A @ x) / nb, flush=True)
cnt[0] = 0
x, info = spla.gmres(A, b, rtol=1e-8, atol=0.0, M=Mop, restart=200, maxiter=1)
print("gmres(200) pairs", cnt[0], "rel", np.linalg.norm(b - A @ x) / nb, flush=True)
cnt[0] = 0
x, info = spla.gmres(A, b, rtol=1e-8, atol=0.0, M=Mop, restart=30, maxiter=20)
print("gmres(30) pairs", cnt[0], "rel", np.linalg.norm(b - A @ x) / nb, flush=True)
for s in (1, 2, 4, 8):
    x, pairs = idrs(A, b, M, s=s)
    print(f"idr({s}) pairs", pairs, "rel", np.linalg.norm(b - A @ x) / nb, flush=True)
cnt[0] = 0
x, info = spla.tfqmr(A, b, rtol=1e-8, atol=0.0, M=Mop, maxiter=300)
print("tfqmr pairs", cnt[0], "rel", np.linalg.norm(b - A @ x) / nb, flush=True)
