"""P1 evaluation of a coarse-mesh field at the nodes of another tet mesh.

The reference warm-starts the fine Navier-Stokes solve from the coarse one with
``create_interpolation_data(..., padding=1e-6)`` + ``interpolate_nonmatching``
(NavierStokesChannelFlow.py:175-194, row a9 of SURVEY 8).  Host-side numpy: a
uniform bucket grid over the coarse tets, then barycentric tests; points outside
every tet (beyond the padding) take the value of the best candidate, clamped.
It only affects the Newton iteration COUNT, never the converged field.
"""
from __future__ import annotations

import numpy as np

from .mesh import TetMesh


def _bary(X, p):
    """Barycentric coordinates of points p (m,3) in tets X (m,4,3) -> (m,4)."""
    T = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2)
    lam = np.linalg.solve(T, (p - X[:, 0])[:, :, None])[:, :, 0]
    return np.concatenate([1.0 - lam.sum(axis=1, keepdims=True), lam], axis=1)


def locate_points(mesh: TetMesh, pts: np.ndarray, padding: float = 1e-6):
    """(tet index, barycentric coords) for every query point."""
    X = mesh.points[mesh.tets]                                   # (E,4,3)
    lo, hi = mesh.points.min(axis=0), mesh.points.max(axis=0)
    ext = np.maximum(hi - lo, 1e-300)
    E = len(mesh.tets)
    res = np.maximum(1, np.round((E / 6.0) ** (1.0 / 3.0) * ext / ext.max() * (ext.max() ** 3 / ext.prod()) ** (1 / 3)))
    res = res.astype(np.int64)
    h = ext / res
    tlo = np.clip(np.floor((X.min(axis=1) - lo) / h - 1e-9).astype(np.int64), 0, res - 1)
    thi = np.clip(np.floor((X.max(axis=1) - lo) / h + 1e-9).astype(np.int64), 0, res - 1)
    span = thi - tlo + 1
    cnt = span.prod(axis=1)
    tid = np.repeat(np.arange(E), cnt)
    off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    s = span[tid]
    k = off % s[:, 2]
    j = (off // s[:, 2]) % s[:, 1]
    i = off // (s[:, 2] * s[:, 1])
    cell = ((tlo[tid, 0] + i) * res[1] + (tlo[tid, 1] + j)) * res[2] + (tlo[tid, 2] + k)
    order = np.argsort(cell, kind="stable")
    cell_s, tid_s = cell[order], tid[order]
    ncell = int(res.prod())
    cptr = np.zeros(ncell + 1, dtype=np.int64)
    np.add.at(cptr, cell_s + 1, 1)
    cptr = np.cumsum(cptr)
    q = np.clip(np.floor((pts - lo) / h).astype(np.int64), 0, res - 1)
    qc = (q[:, 0] * res[1] + q[:, 1]) * res[2] + q[:, 2]
    ncand = cptr[qc + 1] - cptr[qc]
    best_t = np.full(len(pts), -1, dtype=np.int64)
    best_l = np.zeros((len(pts), 4))
    best_m = np.full(len(pts), -np.inf)
    todo = np.arange(len(pts))
    for kk in range(int(ncand.max()) if len(pts) else 0):
        sel = todo[ncand[todo] > kk]
        if len(sel) == 0:
            break
        t = tid_s[cptr[qc[sel]] + kk]
        lam = _bary(X[t], pts[sel])
        mn = lam.min(axis=1)
        better = mn > best_m[sel]
        idx = sel[better]
        best_t[idx], best_l[idx], best_m[idx] = t[better], lam[better], mn[better]
        todo = todo[best_m[todo] < -padding]
    miss = best_t < 0
    if miss.any():                                                # no candidate in the bucket: nearest centroid
        cen = X.mean(axis=1)
        for i0 in np.nonzero(miss)[0]:
            t = int(np.argmin(((cen - pts[i0]) ** 2).sum(axis=1)))
            best_t[i0], best_l[i0] = t, _bary(X[t:t + 1], pts[i0:i0 + 1])[0]
    lam = np.clip(best_l, 0.0, 1.0)
    lam /= lam.sum(axis=1, keepdims=True)
    return best_t, lam


def interpolate_initial_guess(coarse: TetMesh, w_coarse: np.ndarray, fine: TetMesh) -> np.ndarray:
    """Fine-mesh nodal [ux,uy,uz,p] from the coarse solution (:175-194)."""
    t, lam = locate_points(coarse, fine.points)
    Wc = np.asarray(w_coarse).reshape(-1, 4)[coarse.tets[t]]      # (n,4 verts,4 comps)
    return np.einsum("na,nac->nc", lam, Wc).reshape(-1)
