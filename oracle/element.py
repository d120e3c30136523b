"""Closed-form element kernels, vectorised over tets (numpy fp64).

Test infrastructure (see oracle/__init__.py).  Restates, per SURVEY.md
Appendix A, what FFCx generates from
  * setup_stokes_weak_form      NavierStokes/NavierStokesChannelFlow.py:160-172
  * define_navier_stokes_form   :220-251  and its exact Gateaux derivative :46
for P1-P1 on affine tets with the 4-point degree-2 rule (:161,:222).
Local dof order 4*a + c (a = cell-local vertex, c in ux,uy,uz,p); element
matrices are returned as (E,4,4,4,4) = [tet, a, c, b, d] (row (a,c), col (b,d)).
"""
from __future__ import annotations

import numpy as np

QA = 0.1381966011250105
QB = 0.5854101966249685
# phi_a(xi_q) for xi_q in {(a,a,a),(b,a,a),(a,b,a),(a,a,b)}: PHI[q, a]
PHI = np.array([[QB, QA, QA, QA],
                [QA, QB, QA, QA],
                [QA, QA, QB, QA],
                [QA, QA, QA, QB]])
QW = 1.0 / 24.0
GHAT = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
C_I = 36.0                                          # NavierStokesChannelFlow.py:237


def geometry(X):
    """X (E,4,3) -> K (E,3,3)=J^-1, detJ (E,), g (E,4,3) physical P1 gradients."""
    J = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2)   # J[e,i,j]=dx_i/dX_j
    K = np.linalg.inv(J)
    detJ = np.abs(np.linalg.det(J))
    g = np.einsum("ak,ekj->eaj", GHAT, K)
    return K, detJ, g


def cell_diameter(X):
    """UFL CellDiameter: longest vertex-vertex distance (:168)."""
    d = X[:, :, None, :] - X[:, None, :, :]
    return np.sqrt((d * d).sum(-1)).max(axis=(1, 2))


def stokes_element(X):
    """(E,4,4,4,4) Stokes element matrices (:170), mu_T = 0.2 h^2 (:169)."""
    E = X.shape[0]
    _, detJ, g = geometry(X)
    vol = detJ / 6.0
    h = cell_diameter(X)
    mu_T = 0.2 * h * h
    gg = np.einsum("eaj,ebj->eab", g, g)
    A = np.zeros((E, 4, 4, 4, 4))
    for i in range(3):
        A[:, :, i, :, i] = vol[:, None, None] * gg                       # (grad u, grad v)
        A[:, :, i, :, 3] = -(vol / 4.0)[:, None, None] * g[:, :, None, i]   # -(p, div v): int phi_b = vol/4
        A[:, :, 3, :, i] = +(vol / 4.0)[:, None, None] * g[:, None, :, i]   # +(div u, q)
    A[:, :, 3, :, 3] = (mu_T * vol)[:, None, None] * gg                  # mu_T (grad p, grad q)
    return A


def ns_element(X, W, Re, *, want_jac: bool = True):
    """NS residual (E,4,4) [tet,a,c] and exact Jacobian (E,4,4,4,4) [tet,a,c,b,d].

    X (E,4,3) vertex coords in cell-local order, W (E,4,4) nodal [ux,uy,uz,p].
    """
    E = X.shape[0]
    nu = 1.0 / Re                                                        # :223
    K, detJ, g = geometry(X)
    G = np.einsum("eki,ekj->eij", K, K)                                  # G = K^T K  :235
    trG = np.einsum("eii->e", G)
    GG = (G * G).sum(axis=(1, 2))
    U, P = W[:, :, :3], W[:, :, 3]
    gu = np.einsum("eai,eaj->eij", U, g)                                 # grad(u)[i,j]
    divu = np.einsum("eii->e", gu)
    gp = np.einsum("ea,eaj->ej", P, g)
    gab = np.einsum("eaj,ebj->eab", g, g)                                # g_a . g_b
    gu_ga = np.einsum("eij,eaj->eai", gu, g)                             # (grad u) g_a  [a,i]
    R = np.zeros((E, 4, 4))
    Jm = np.zeros((E, 4, 4, 4, 4)) if want_jac else None
    I3 = np.eye(3)
    for q in range(4):
        phi = PHI[q]
        wd = QW * detJ
        u = np.einsum("a,eai->ei", phi, U)
        p = np.einsum("a,ea->e", phi, P)
        Gu = np.einsum("eij,ej->ei", G, u)
        tau = 1.0 / np.sqrt(np.einsum("ei,ei->e", u, Gu) + C_I * nu * nu * GG)    # :238
        nuL = 1.0 / (trG * tau)                                                   # :249
        conv = np.einsum("eij,ej->ei", gu, u)                            # (u . nabla) u      :243
        r = np.einsum("eij,ei->ej", gu, u) + gp                          # dot(u,grad u)+grad p :241
        s = np.einsum("ej,eaj->ea", r, g)                                # r . g_a
        # ---- residual ----
        Rm = (conv[:, None, :] * phi[None, :, None]
              + nu * gu_ga
              - p[:, None, None] * g
              + (tau[:, None] * s)[:, :, None] * u[:, None, :]
              + (nuL * divu)[:, None, None] * g)
        Rc = phi[None, :] * divu[:, None] + tau[:, None] * s
        R[:, :, :3] += wd[:, None, None] * Rm
        R[:, :, 3] += wd[:, None] * Rc
        if not want_jac:
            continue
        # ---- Jacobian: direction (b,j): du = phi_b e_j, d(grad u) = e_j (x) g_b ----
        ugb = np.einsum("ej,ebj->eb", u, g)                              # g_b . u
        dtau = -(tau ** 3)[:, None, None] * phi[None, :, None] * Gu[:, None, :]      # [b,j]
        dnuL = (tau / trG)[:, None, None] * phi[None, :, None] * Gu[:, None, :]      # [b,j]
        # d r . g_a  = u_j (g_b.g_a) + phi_b (grad u)[j,:].g_a      -> [a,b,j]
        drga = u[:, None, None, :] * gab[:, :, :, None] + phi[None, None, :, None] * gu_ga[:, :, None, :]
        # momentum rows (a,i), velocity cols (b,j)
        Juu = np.zeros((E, 4, 3, 4, 3))
        # convection phi_a [ delta_ij (g_b.u) + gu_ij phi_b ]
        Juu += phi[None, :, None, None, None] * (
            I3[None, None, :, None, :] * ugb[:, None, None, :, None]
            + gu[:, None, :, None, :] * phi[None, None, None, :, None])
        # viscous nu delta_ij g_a.g_b
        Juu += nu * I3[None, None, :, None, :] * gab[:, :, None, :, None]
        # SUPG: dtau u_i s_a + tau delta_ij phi_b s_a + tau u_i (dr . g_a)
        Juu += dtau[:, None, None, :, :] * (u[:, None, :, None, None] * s[:, :, None, None, None])
        Juu += (tau[:, None] * s)[:, :, None, None, None] * I3[None, None, :, None, :] * phi[None, None, None, :, None]
        Juu += tau[:, None, None, None, None] * u[:, None, :, None, None] * drga[:, :, None, :, :]
        # LSIC: dnuL divu g_a[i] + nuL g_b[j] g_a[i]
        Juu += dnuL[:, None, None, :, :] * (divu[:, None, None] * g)[:, :, :, None, None]
        Juu += nuL[:, None, None, None, None] * g[:, :, :, None, None] * g[:, None, None, :, :]
        # momentum rows, pressure cols: -phi_b g_a[i] + tau u_i g_a.g_b
        Jup = (-phi[None, None, None, :] * g[:, :, :, None]
               + tau[:, None, None, None] * u[:, None, :, None] * gab[:, :, None, :])
        # continuity rows (a,p), velocity cols: phi_a g_b[j] + dtau s_a + tau dr.g_a
        Jpu = (phi[None, :, None, None] * g[:, None, :, :]
               + dtau[:, None, :, :] * s[:, :, None, None]
               + tau[:, None, None, None] * drga)
        Jpp = tau[:, None, None] * gab
        w5 = wd[:, None, None, None, None]
        Jm[:, :, :3, :, :3] += w5 * Juu
        Jm[:, :, :3, :, 3] += wd[:, None, None, None] * Jup
        Jm[:, :, 3, :, :3] += wd[:, None, None, None] * Jpu
        Jm[:, :, 3, :, 3] += wd[:, None, None] * Jpp
    return R, Jm
