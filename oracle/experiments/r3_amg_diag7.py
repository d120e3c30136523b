"""Aggregate-block Jacobi as fine-level smoother: blocks = the nodes of each level-0 aggregate (32 x 32 for 8 nodes), or of a SHIFTED
aggregation (seeds taken in reverse node order) so that pre- and post-smoothing use different block boundaries."""
import sys, time, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.proto_amg import problem
from oracle.proto_sa import setup, aggregate, describe
from stabilized_navier_stokes_flow_fenicsx_amd import _lib

def block_inverse(A, agg, nc):
    """LU of the diagonal blocks of A over the aggregates, as one block-diagonal sparse matrix in a permuted numbering"""
    n = A.shape[0] // 4
    order = np.argsort(agg, kind="stable")
    dofs = (4 * order[:, None] + np.arange(4)[None]).ravel()
    Ap = A[dofs][:, dofs].tocsr()
    sizes = 4 * np.bincount(agg, minlength=nc)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    blocks = []
    for I in range(nc):
        blocks.append(np.linalg.inv(Ap[ptr[I]:ptr[I + 1], ptr[I]:ptr[I + 1]].toarray()))
    Binv_p = sp.block_diag(blocks, format="csr")
    Pm = sp.csr_matrix((np.ones(len(dofs)), (np.arange(len(dofs)), dofs)), shape=A.shape)   # permuted = Pm @ original
    return (Pm.T @ Binv_p @ Pm).tocsr()

cells = tuple(int(a) for a in sys.argv[1:4]); Re = float(sys.argv[4])
A, b, free = problem(cells, Re)
print("dofs", A.shape[0], "Re", Re, flush=True)
lv = setup(A, free)
L = lv[0]
n = L.n
Ab = L.A.tobsr((4, 4)); Ab.sort_indices()
agg, nc = _lib.host_aggregate(Ab.indptr, Ab.indices, None, 8)
# shifted aggregation: reverse the node numbering
perm = np.arange(n)[::-1]
Ar = sp.csr_matrix((np.ones(len(Ab.indices)), Ab.indices, Ab.indptr), shape=(n, n))[perm][:, perm].tocsr(); Ar.sort_indices()
agg2r, nc2 = _lib.host_aggregate(Ar.indptr.astype(np.int32), Ar.indices.astype(np.int32), None, 8)
agg2 = np.empty(n, dtype=agg2r.dtype); agg2[perm] = agg2r
t0 = time.time(); B1 = block_inverse(L.A, agg, nc); B2 = block_inverse(L.A, agg2, nc2); print(f"block inverses {time.time() - t0:.0f}s", flush=True)

def lam(Binv):
    rng = np.random.default_rng(0); x = rng.normal(size=A.shape[0]); l = 1
    for _ in range(15):
        y = Binv @ (L.A @ x); l = np.linalg.norm(y) / np.linalg.norm(x); x = y / np.linalg.norm(y)
    return l
l1, l2 = lam(B1), lam(B2)
print("lambda_max(Binv A) aggregate blocks", l1, "shifted", l2, "point-block", L.lam, flush=True)

from oracle.proto_sa import cycle as plain_cycle
def vcycle(pre, post):
    def f(v):
        x = pre(np.zeros_like(v), v)
        r = v - L.A @ x
        xc = plain_cycle(lv, 1, L.R @ r, (1, 4, 6, 2), (1, 4, 6, 2))
        x = x + L.P @ xc
        return post(x, v)
    return f
def sm(Binv, om, nu=1):
    def g(x, v):
        for _ in range(nu):
            x = x + om * (Binv @ (v - L.A @ x))
        return x
    return g
def run(label, f):
    its = [0]
    M_ = spla.LinearOperator(A.shape, matvec=f)
    x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=M_, maxiter=300, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    print(f"{label:70s} its {its[0]} info {info} rel {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.1e}", flush=True)
pj = sm(L.Dinv, L.omega)
run("V(1,1) point-block Jacobi (the product)", vcycle(pj, pj))
run("V(2,2) point-block Jacobi", vcycle(sm(L.Dinv, L.omega, 2), sm(L.Dinv, L.omega, 2)))
for om_f in (1.0, 0.8):
    o1, o2 = min(0.9, om_f * 4 / (3 * l1)), min(0.9, om_f * 4 / (3 * l2))
    run(f"V(1,1) aggregate-block Jacobi, same blocks pre+post (omega {o1:.2f})", vcycle(sm(B1, o1), sm(B1, o1)))
    run(f"V(1,1) aggregate blocks pre, SHIFTED blocks post (omega {o1:.2f}/{o2:.2f})", vcycle(sm(B1, o1), sm(B2, o2)))
    run(f"V(1,1) shifted blocks pre and post (omega {o2:.2f})", vcycle(sm(B2, o2), sm(B2, o2)))
    run(f"V(1,1) point-block pre, shifted aggregate blocks post", vcycle(pj, sm(B2, o2)))
