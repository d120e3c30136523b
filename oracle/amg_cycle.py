"""CPU restatement (numpy / scipy) of the AMG V-cycle libsns applies as preconditioner -- TEST INFRASTRUCTURE like the rest of
oracle/: imported by tests/ only, never by the product (the product path is the HIP kernels of csrc/; the reference itself
preconditions with PETSc's ILU(0), NavierStokesChannelFlow.py:274-283, so there is no reference code to follow here: this file
restates the PRODUCT's own algorithm, DESIGN.md section 4, so that the GPU cycle can be checked operator-for-operator).

What is restated (single GPU):
  * hierarchy: greedy aggregates of <= 8 nodes from the product's host utility (sns_host_aggregate -- symbolic, CPU), 4 dofs per
    aggregate, Galerkin operators P^T A P with the level-0 Dirichlet dofs excluded from the transfer and a unit diagonal on
    empty coarse dofs; coarsening stops at the first level >= 1 with <= max(coarse_nodes, dense_rows) rows, which is solved exactly;
  * smoothers: damped nodal-block Jacobi x <- x + w D^-1 (b - A x) (D = the 4 x 4 diagonal blocks), or -- on the levels listed
    in `block_levels` -- aggregate-block Jacobi with B = the diagonal blocks of A over the aggregates that form the next level;
  * cycle: first pre-sweep from the zero guess (w S b), nu_pre - 1 further sweeps, residual, restriction P^T, recursive coarse
    solve, correction, nu_post sweeps.  Damping per level is an INPUT (the GPU's own values, FlowProblem.hierarchy()), because
    the estimate is not part of the operator being compared.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from stabilized_navier_stokes_flow_fenicsx_amd import _lib      # host-only symbolic utility (no GPU needed)


class Level:
    pass


def nodal_block_inverse(A, n):
    Ab = A.tobsr((4, 4))
    Ab.sort_indices()
    rows = np.repeat(np.arange(n), np.diff(Ab.indptr))
    sel = Ab.indices == rows
    D = np.zeros((n, 4, 4))
    D[rows[sel]] = Ab.data[sel]
    return sp.bsr_matrix((np.linalg.inv(D), np.arange(n), np.arange(n + 1)), shape=(4 * n, 4 * n)).tocsr()


def smoother_blocks(agg, nc):
    """the product's smoother blocks: the aggregates, one of more than 8 nodes split into chunks of 8 in ascending node order"""
    blk = np.empty_like(agg)
    nb = 0
    order = np.argsort(agg, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(agg, minlength=nc))])
    for I in range(nc):
        mem = order[ptr[I]:ptr[I + 1]]
        for k in range(0, len(mem), 8):
            blk[mem[k:k + 8]] = nb
            nb += 1
    return blk, nb


def aggregate_block_inverse(A, agg, nc):
    """blockdiag over the smoother blocks (aggregates, see smoother_blocks) of A, inverted, in the original numbering"""
    agg, nc = smoother_blocks(agg, nc)
    order = np.argsort(agg, kind="stable")
    dofs = (4 * order[:, None] + np.arange(4)[None]).ravel()
    Ap = A[dofs][:, dofs].tocsr()
    ptr = np.concatenate([[0], np.cumsum(4 * np.bincount(agg, minlength=nc))])
    blocks = [np.linalg.inv(Ap[ptr[i]:ptr[i + 1], ptr[i]:ptr[i + 1]].toarray()) for i in range(nc)]
    Bp = sp.block_diag(blocks, format="csr")
    Pm = sp.csr_matrix((np.ones(len(dofs)), (np.arange(len(dofs)), dofs)), shape=A.shape)
    return (Pm.T @ Bp @ Pm).tocsr()


def node_graph(A):
    """structural node graph (n x n, ones) of a matrix with 4 dofs per node: one entry per stored 4 x 4 block"""
    Ab = A.tobsr((4, 4))
    Ab.sort_indices()
    n = A.shape[0] // 4
    return sp.csr_matrix((np.ones(len(Ab.indices)), Ab.indices.copy(), Ab.indptr.copy()), shape=(n, n))


def build(A, free, coarse_nodes=32, dense_rows=512, agg_size=8, max_levels=12, block_levels=(), graph=None):
    """levels of the product's serial hierarchy for the fine operator A (scipy sparse, 4 dofs per node) and its free-dof mask.
    `graph`: the STRUCTURAL node graph of A (the BSR pattern the product assembles into, explicit zero blocks included; default:
    A's stored blocks).  The product aggregates on patterns, not values: a coarse pattern is the image of the fine one, whatever
    cancels numerically (Dirichlet rows / columns are zeroed in the values but stay in the pattern)."""
    levels = []
    stop = max(coarse_nodes, min(dense_rows, 4096))
    G = node_graph(A) if graph is None else graph.tocsr()
    while True:
        L = Level()
        n = A.shape[0] // 4
        L.A, L.n = A.tocsr(), n
        L.S = nodal_block_inverse(L.A, n)
        L.P = None
        levels.append(L)
        l = len(levels) - 1
        if n <= coarse_nodes or (l >= 1 and n <= stop) or len(levels) >= max_levels:
            break
        G.sort_indices()
        agg, nc = _lib.host_aggregate(G.indptr.astype(np.int32), G.indices.astype(np.int32), None, agg_size)
        if nc >= n or nc == 0:
            break
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        Gc = (T.T @ G @ T).tocsr()                   # all-positive data: nothing cancels, the pattern is the image of G
        Gc.data[:] = 1.0
        dof = np.arange(4 * n)
        col = 4 * agg[dof // 4].astype(np.int64) + dof % 4
        w = np.ones(4 * n) if free is None else np.asarray(free, dtype=np.float64)
        L.P = sp.csr_matrix((w, (dof, col)), shape=(4 * n, 4 * nc))
        Ac = (L.P.T @ L.A @ L.P).tocsr()
        empty = np.asarray(abs(Ac).sum(axis=1)).ravel() == 0
        if empty.any():
            Ac = Ac + sp.diags(empty.astype(np.float64))
        if l in block_levels:
            L.S = aggregate_block_inverse(L.A, agg, nc)
        A, free, G = Ac, None, Gc
    last = levels[-1]
    last.exact = last.n <= max(stop, 40) and len(levels) > 1
    if last.exact:
        last.lu = spla.splu(sp.csc_matrix(last.A))
    return levels


def cycle(levels, l, b, sweeps, omega):
    """x ~ A_l^-1 b.  sweeps[l] = (nu_pre, nu_post), omega[l] = damping of level l"""
    L = levels[l]
    om = omega[l]
    if L.P is None:
        if L.exact:
            return L.lu.solve(b)
        x = om * (L.S @ b)                       # a last level too large for the direct solve: 1 + 8 sweeps
        for _ in range(8):
            x = x + om * (L.S @ (b - L.A @ x))
        return x
    nu_pre, nu_post = sweeps[l]
    x = om * (L.S @ b)
    for _ in range(nu_pre - 1):
        x = x + om * (L.S @ (b - L.A @ x))
    xc = cycle(levels, l + 1, L.P.T @ (b - L.A @ x), sweeps, omega)
    x = x + L.P @ xc
    for _ in range(nu_post):
        x = x + om * (L.S @ (b - L.A @ x))
    return x
