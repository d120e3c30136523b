"""Development prototype (CPU, scipy): the aggregation-AMG preconditioner of libsns restated with sparse
matrices, to try cycle variants (V / W / K) offline before they are written as HIP.  Test infrastructure
like the rest of oracle/: never imported by the product.

    python -m oracle.proto_amg 96 24 24 64      # duct cells, Re
"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import cport
from stabilized_navier_stokes_flow_fenicsx_amd import _lib, bcs as B, mesh as M


class Level:
    pass


def block_diag_inv(A, n):
    Ab = A.tobsr((4, 4))
    Ab.sort_indices()
    D = np.zeros((n, 4, 4))
    for i in range(n):
        pass
    rows = np.repeat(np.arange(n), np.diff(Ab.indptr))
    sel = Ab.indices == rows
    D[rows[sel]] = Ab.data[sel]
    Di = np.linalg.inv(D)
    return sp.bsr_matrix((Di, np.arange(n), np.arange(n + 1)), shape=(4 * n, 4 * n)).tocsr()


def lam_max(A, Dinv, its=12):
    rng = np.random.default_rng(0)
    x = rng.normal(size=A.shape[0])
    lam = 1.0
    for _ in range(its):
        y = Dinv @ (A @ x)
        lam = np.linalg.norm(y) / np.linalg.norm(x)
        x = y / np.linalg.norm(y)
    return lam


def setup(A, free, coarse_size=256, max_levels=12, agg_size=8):
    levels = []
    while True:
        L = Level()
        n = A.shape[0] // 4
        L.A, L.n = A.tocsr(), n
        L.Dinv = block_diag_inv(A, n)
        lam = lam_max(L.A, L.Dinv)
        L.omega = min(0.8, 4.0 / (3.0 * lam))
        levels.append(L)
        if n <= coarse_size or len(levels) >= max_levels:
            L.lu = spla.splu(sp.csc_matrix(L.A))
            break
        Ab = L.A.tobsr((4, 4))
        Ab.sort_indices()
        agg, nc = _lib.host_aggregate(Ab.indptr, Ab.indices, None, agg_size)
        dof = np.arange(4 * n)
        col = 4 * agg[dof // 4].astype(np.int64) + dof % 4
        w = np.ones(4 * n) if free is None else free.astype(np.float64)
        P = sp.csr_matrix((w, (dof, col)), shape=(4 * n, 4 * nc))
        Ac = (P.T @ L.A @ P).tocsr()
        d = Ac.diagonal()
        empty = np.asarray(abs(Ac).sum(axis=1)).ravel() == 0
        if empty.any():
            Ac = Ac + sp.diags(empty.astype(np.float64))
        L.P = P
        A, free = Ac, None
    return levels


def smooth(L, x, b, nu):
    for _ in range(nu):
        x = x + L.omega * (L.Dinv @ (b - L.A @ x))
    return x


def nu_of(l, sched):
    return sched[min(l, len(sched) - 1)]


def cycle(levels, l, b, sched=(1, 4, 6, 2), kind="V", klevels=(1, 2), counter=None):
    L = levels[l]
    if counter is not None:
        counter[l] = counter.get(l, 0) + 1
    if l == len(levels) - 1:
        return L.lu.solve(b)
    nu = nu_of(l, sched)
    x = L.omega * (L.Dinv @ b)
    x = smooth(L, x, b, nu - 1)
    r = b - L.A @ x
    bc = L.P.T @ r
    C = levels[l + 1]
    rec = lambda v: cycle(levels, l + 1, v, sched, kind, klevels, counter)
    if kind == "K" and (l + 1) in klevels and l + 1 < len(levels) - 1:
        # two steps of GCR on A_c xc = bc, preconditioned by the next-level cycle (Notay's K-cycle, nonsymmetric form)
        c1 = rec(bc)
        v1 = C.A @ c1
        n1 = v1 @ v1
        a1 = (v1 @ bc) / n1
        r1 = bc - a1 * v1
        if np.linalg.norm(r1) <= 0.25 * np.linalg.norm(bc):
            xc = a1 * c1
        else:
            c2 = rec(r1)
            v2 = C.A @ c2
            g = (v2 @ v1) / n1
            c2 = c2 - g * c1
            v2 = v2 - g * v1
            a2 = (v2 @ r1) / (v2 @ v2)
            xc = a1 * c1 + a2 * c2
    elif kind == "W" and (l + 1) in klevels and l + 1 < len(levels) - 1:
        xc = rec(bc)
        xc = xc + rec(bc - C.A @ xc)
    else:
        xc = rec(bc)
    x = x + L.P @ xc
    return smooth(L, x, b, nu)


def run(A, b, levels, label, **kw):
    cnt = {}
    M_ = spla.LinearOperator(A.shape, matvec=lambda v: cycle(levels, 0, v, counter=cnt, **kw))
    its = [0]
    t0 = time.time()
    x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=M_, maxiter=400, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    rel = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    print(f"{label:34s} its {its[0]:4d} info {info} rel {rel:.1e} visits/level {[cnt.get(l, 0) for l in range(len(levels))]} "
          f"{time.time() - t0:.1f}s", flush=True)
    return its[0]


def problem(cells, Re, length=4.0):
    m = M.duct_mesh(cells, length)
    mask, g = B.duct_bcs(m).flatten()
    x, y, z = m.points.T
    w = np.zeros(m.num_dofs)
    w[0::4] = 2.25 * (1 - 4 * y * y) * (1 - 4 * z * z)
    w[3::4] = 30.0 / Re * (length - x)
    Bm = mask.astype(bool)
    w[Bm] = g[Bm]
    rp, ci = cport.pattern(m.num_nodes, m.tets)
    vals, F = cport.assemble("ns", m.points, m.tets, w, Re, mask, g, rp, ci)
    A = cport.to_scipy(m.num_nodes, rp, ci, vals)
    return A, -F, ~Bm


if __name__ == "__main__":
    cells = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 16, 16)
    Re = float(sys.argv[4]) if len(sys.argv) > 4 else 200.0 * cells[1] / 75.0
    length = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
    A, b, free = problem(cells, Re, length)
    t0 = time.time()
    levels = setup(A, free)
    print("cells", cells, "Re", Re, "dofs", A.shape[0], "levels", [L.n for L in levels], "omega",
          [round(L.omega, 2) for L in levels], f"setup {time.time() - t0:.1f}s", flush=True)
    run(A, b, levels, "V (1,4,6,2)")
    run(A, b, levels, "W at 1,2", kind="W")
    run(A, b, levels, "K at 1,2", kind="K")
    run(A, b, levels, "K at 1", kind="K", klevels=(1,))
    run(A, b, levels, "K at 1,2,3", kind="K", klevels=(1, 2, 3))
    run(A, b, levels, "K at 1..5 (1,2,2,2)", kind="K", klevels=(1, 2, 3, 4, 5), sched=(1, 2, 2, 2))
