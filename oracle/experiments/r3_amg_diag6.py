"""Non-Galerkin level 1: sparsify the level-1 operator (drop blocks whose Frobenius norm is below theta x the row's largest
off-diagonal block, lump them onto the diagonal block) and build the deeper levels from the sparsified operator."""
import sys, time, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.proto_amg import problem, Level, block_diag_inv, lam_max
from oracle.proto_sa import aggregate, cycle, run, describe

def sparsify(A, theta, lump="diag"):
    Ab = A.tobsr((4, 4)); Ab.sort_indices()
    n = Ab.shape[0] // 4
    rows = np.repeat(np.arange(n), np.diff(Ab.indptr))
    nrm = np.sqrt((Ab.data ** 2).sum(axis=(1, 2)))
    off = Ab.indices != rows
    mx = np.zeros(n); np.maximum.at(mx, rows[off], nrm[off])
    keep = (~off) | (nrm >= theta * mx[rows])
    data = Ab.data.copy()
    if lump == "diag":
        dsl = np.zeros(n, dtype=np.int64); dsl[rows[~off]] = np.nonzero(~off)[0]
        np.add.at(data, dsl[rows[~keep]], Ab.data[~keep])
    ind = Ab.indices[keep]; dat = data[keep]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))])
    return sp.bsr_matrix((dat, ind, ptr), shape=Ab.shape).tocsr(), keep.sum() / len(keep)

def setup(A, free, theta, lump, sparsify_levels=(1,), coarse_size=256):
    levels = []; l = 0
    while True:
        L = Level(); n = A.shape[0] // 4
        if l in sparsify_levels and theta > 0:
            A, frac = sparsify(A, theta, lump)
        L.A, L.n = A.tocsr(), n
        L.nnzb = L.A.tobsr((4, 4)).indices.size
        L.Dinv = block_diag_inv(A, n); L.lam = lam_max(L.A, L.Dinv); L.omega = min(0.8, 4.0 / (3.0 * L.lam))
        levels.append(L)
        if n <= coarse_size or len(levels) >= 12:
            L.lu = spla.splu(sp.csc_matrix(L.A)); break
        agg, nc = aggregate(L.A, n, 8)
        dof = np.arange(4 * n); col = 4 * agg[dof // 4].astype(np.int64) + dof % 4
        w = np.ones(4 * n) if free is None else free.astype(np.float64)
        P = sp.csr_matrix((w, (dof, col)), shape=(4 * n, 4 * nc))
        Ac = (P.T @ L.A @ P).tocsr()
        empty = np.asarray(abs(Ac).sum(axis=1)).ravel() == 0
        if empty.any(): Ac = Ac + sp.diags(empty.astype(np.float64))
        L.P, L.R = P, P.T.tocsr(); L.pnnzb = n
        A, free = Ac, None; l += 1
    return levels

cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
print("dofs", A.shape[0], "Re", Re, flush=True)
for theta, lump, lv_s in [(0.0, "diag", (1,)), (0.1, "diag", (1,)), (0.2, "diag", (1,)), (0.3, "diag", (1,)), (0.2, "none", (1,)), (0.2, "diag", (1, 2)), (0.3, "diag", (1, 2, 3))]:
    lv = setup(A, free, theta, lump, lv_s)
    print(f"theta {theta} lump {lump} levels {lv_s}:", describe(lv), flush=True)
    run(A, b, lv, f"  V(1,4,6,2)", (1, 4, 6, 2), (1, 4, 6, 2))
    run(A, b, lv, f"  V pre(1,1,6,2) post(1,6,6,2)", (1, 1, 6, 2), (1, 6, 6, 2))
