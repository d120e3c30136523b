"""Boundary functionals of a P1-P1 solution: traction force on tagged facets, drag / lift coefficients.

Mirrors the post-processing of the reference's DFG benchmark script
(Validation_Flow/DFG_3D_Validation.py:344-367):

    n        = -FacetNormal(msh)                       # pointing out of the obstacle, into the fluid
    stress   = -p I + 2 mu sym(grad u)
    traction = stress . n
    F_drag   = assemble(traction[0] * ds(obstacle)),  F_lift = assemble(traction[1] * ds(obstacle))
    C_d      = 2 F_drag / (rho Uc^2 Lc),               C_l   = 2 F_lift / (rho Uc^2 Lc)

For P1 fields grad u is constant in the tet behind a boundary facet and p is linear on the facet, so the
facet integrals are exact with  area * (stress(grad u, mean of the 3 nodal p) . n).  Host side (numpy): the
obstacle surface holds O(N^(2/3)) facets, there is nothing to accelerate.
"""
from __future__ import annotations

import numpy as np

from .mesh import TetMesh


def facet_parent_tets(mesh: TetMesh, facet_ids: np.ndarray) -> np.ndarray:
    """Index of the (single) tet behind each boundary facet."""
    f = np.sort(mesh.facets[facet_ids].astype(np.int64), axis=1)
    n = mesh.num_nodes
    want = (f[:, 0] * n + f[:, 1]) * n + f[:, 2]
    t = mesh.tets.astype(np.int64)
    faces = np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]])
    faces.sort(axis=1)
    key = (faces[:, 0] * n + faces[:, 1]) * n + faces[:, 2]
    order = np.argsort(key, kind="stable")
    pos = np.searchsorted(key[order], want)
    if np.any(pos >= len(key)) or np.any(key[order][np.minimum(pos, len(key) - 1)] != want):
        raise ValueError("a boundary facet is not a face of any tet")
    return (order[pos] % len(t)).astype(np.int64)


def boundary_traction_force(mesh: TetMesh, w: np.ndarray, nu: float, tag: int) -> np.ndarray:
    """int_{facets tagged ``tag``} (-p I + 2 nu sym grad u) . n ds  with  n = -(outward normal of the fluid domain),
    as a 3-vector (DFG_3D_Validation.py:348-356: drag = component 0, lift = component 1)."""
    ids = mesh.find(tag)
    if len(ids) == 0:
        return np.zeros(3)
    W = np.asarray(w, dtype=np.float64).reshape(-1, 4)
    par = facet_parent_tets(mesh, ids)
    tn = mesh.tets[par].astype(np.int64)                     # (F,4)
    X = mesh.points[tn]                                      # (F,4,3)
    J = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2)       # J_ij = dx_i/dX_j
    K = np.linalg.inv(J)                                     # K_ji = dX_j/dx_i
    g = np.concatenate([-K.sum(axis=1, keepdims=True), K], axis=1)                      # (F,4,3) grad phi_a
    gu = np.einsum("fai,faj->fij", W[tn][:, :, :3], g)       # (grad u)_ij = d u_i / d x_j
    fn = mesh.facets[ids].astype(np.int64)
    P = mesh.points[fn]                                      # (F,3,3)
    cr = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])      # |cr| = 2 area
    # orient outward from the fluid: away from the tet's 4th (non-facet) vertex
    opp = tn.sum(axis=1) - fn.sum(axis=1)                    # the facet's 3 nodes are 3 of the tet's 4
    sgn = np.sign(np.einsum("fi,fi->f", cr, P[:, 0] - mesh.points[opp]))
    n_area = -0.5 * cr * sgn[:, None]                        # n ds with n = -outward
    pm = W[fn][:, :, 3].mean(axis=1)
    stress = 2.0 * nu * 0.5 * (gu + gu.transpose(0, 2, 1))
    stress[:, [0, 1, 2], [0, 1, 2]] -= pm[:, None]
    return np.einsum("fij,fj->i", stress, n_area)


def drag_lift_coefficients(force: np.ndarray, rho: float = 1.0, Uc: float = 0.2, Lc: float = 0.1 * 0.41):
    """(C_d, C_l) = 2 F / (rho Uc^2 Lc)  (DFG_3D_Validation.py:345-346,364-365; defaults are the script's)."""
    s = 2.0 / (rho * Uc * Uc * Lc)
    return s * float(force[0]), s * float(force[1])
