"""Literal, term-by-term evaluation of the reference's UFL forms on ONE tet.

Test infrastructure (see oracle/__init__.py).  Slow on purpose: every UFL
operator of NavierStokes/NavierStokesChannelFlow.py:160-172 and :220-251 is
spelled out with explicit test functions so that the closed forms in
``oracle/element.py`` can be checked against the text of the weak form rather
than against a second hand derivation.  The Jacobian is the automatic
derivative of the residual w.r.t. the 16 nodal coefficients, i.e. the Gateaux
derivative ``ufl.derivative(F, w, dw)`` of :46 / :253-254.

UFL conventions used (fenics-ufl 2024.2.0):
  grad(f)[..., j] = d f[...] / d x_j          nabla_grad(f)[j, ...] = d f[...] / d x_j
  dot(a, b) contracts last index of a with first of b
  inner = full contraction          div(v) = sum_i d v_i / d x_i
  A * B (rank-2) = matrix product   tr, inv as usual
  Jacobian(mesh)[i, j] = d x_i / d X_j with reference tet (0,0,0),(1,0,0),(0,1,0),(0,0,1)
  dx(degree=2) on a tetrahedron -> 4-point rule, weights 1/24 (basix default)
Local dof order: 4*a + c, a = cell-local vertex, c in (ux, uy, uz, p).
"""
from __future__ import annotations

import numpy as np
import torch

QA = 0.1381966011250105
QB = 0.5854101966249685
QPTS = np.array([[QA, QA, QA], [QB, QA, QA], [QA, QB, QA], [QA, QA, QB]])
QW = np.full(4, 1.0 / 24.0)
GHAT = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])

_T = torch.float64

# Perturbations of the stabilisation for the study of what the reference-held constants tell apart (round 5, DESIGN.md section 5;
# the product's sns_set_form_variant): C_I of :237, a factor on the LSIC coefficient of :249, the sign of the PSPG part of the test
# function of :247, and all four quadrature points moved to the centroid (a 1-point rule).  The defaults ARE the reference's form.
VARIANT = dict(ci=36.0, lsic=1.0, pspg=1.0, one_point=False)


def _phi(xi):
    return torch.stack([1.0 - xi[0] - xi[1] - xi[2], xi[0], xi[1], xi[2]])


def _geometry(X):
    """X (4,3) -> Jacobian, inverse, |det|, physical basis gradients (4,3)."""
    J = torch.stack([X[1] - X[0], X[2] - X[0], X[3] - X[0]], dim=1)     # J[i,j] = dx_i/dX_j
    K = torch.linalg.inv(J)                                              # dX/dx
    detJ = torch.abs(torch.linalg.det(J))
    gphi = torch.as_tensor(GHAT, dtype=_T) @ K                           # d phi_a / d x_j
    return J, K, detJ, gphi


def ns_residual_literal(X, w, Re, *, corrected_convection: bool = False):
    """16-vector F(w; v_a e_i, q_a) of NavierStokesChannelFlow.py:243-251 for one tet.

    X: (4,3) vertices, w: (16,) nodal [ux,uy,uz,p]*4 (torch fp64, may require grad).
    ``corrected_convection`` swaps the reference's dot(u, grad(.)) for the
    convective derivative in res_M and in the SUPG test function;
    default False = the reference as written.
    """
    X = torch.as_tensor(X, dtype=_T)
    nu = 1.0 / Re                                                        # :223
    _, K, detJ, gphi = _geometry(X)
    W = w.reshape(4, 4)
    U, P = W[:, :3], W[:, 3]
    # metric tensor :232-235 ; grad(x) = I
    dxi_dy = K
    dxi_dx = dxi_dy @ torch.linalg.inv(torch.eye(3, dtype=_T))
    G = dxi_dx.T @ dxi_dx
    Ci = VARIANT["ci"]                                                    # :237 (36)
    grad_u = torch.einsum("ai,aj->ij", U, gphi)                          # grad(u)[i,j] = du_i/dx_j
    nabla_grad_u = grad_u.T
    div_u = torch.trace(grad_u)
    grad_p = torch.einsum("a,aj->j", P, gphi)
    I3 = torch.eye(3, dtype=_T)
    out = torch.zeros(16, dtype=_T)
    res = []
    for q in range(4):
        phi = _phi(torch.as_tensor([0.25, 0.25, 0.25] if VARIANT["one_point"] else QPTS[q], dtype=_T))
        u = torch.einsum("a,ai->i", phi, U)
        p = torch.dot(phi, P)
        tau = 1.0 / torch.sqrt(torch.dot(u, G @ u) + Ci * nu ** 2 * torch.sum(G * G))   # :238
        # sigma = 2 nu sym(grad u) - p I ; P1 on an affine cell: second derivatives vanish,
        # so div(sigma) = -grad(p)                                                       :240
        div_sigma = -grad_p
        if corrected_convection:
            res_M = u @ nabla_grad_u - div_sigma
        else:
            res_M = u @ grad_u - div_sigma                               # dot(u, grad(u)) :241
        v_lsic = VARIANT["lsic"] / (torch.trace(G) * tau)                # :249
        wq = QW[q] * detJ
        vals = []
        for a in range(4):
            for c in range(4):
                if c < 3:                                                # test (v, q) = (phi_a e_c, 0)
                    v = phi[a] * I3[c]
                    grad_v = torch.outer(I3[c], gphi[a])                 # grad(v)[i,j] = dv_i/dx_j
                    div_v = gphi[a, c]
                    qt = torch.zeros((), dtype=_T)
                    grad_q = torch.zeros(3, dtype=_T)
                else:                                                    # test (0, phi_a)
                    v = torch.zeros(3, dtype=_T)
                    grad_v = torch.zeros(3, 3, dtype=_T)
                    div_v = torch.zeros((), dtype=_T)
                    qt = phi[a]
                    grad_q = gphi[a]
                t = torch.dot(u @ nabla_grad_u, v)                       # :243
                t = t + nu * torch.sum(grad_u * grad_v)                  # :244
                t = t - p * div_v                                        # :245
                t = t + qt * div_u                                       # :246
                supg_test = (u @ nabla_grad(grad_v) if corrected_convection else u @ grad_v) + VARIANT["pspg"] * grad_q
                t = t + torch.dot(tau * res_M, supg_test)                # :247
                t = t + v_lsic * div_v * div_u                           # :251
                vals.append(wq * t)
        res.append(torch.stack(vals))
    out = torch.stack(res).sum(dim=0)
    return out


def nabla_grad(g):
    return g.T


def ns_residual_and_jacobian_literal(X, w, Re, **kw):
    """(F (16,), J (16,16)) with J = dF/dw by reverse-mode autodiff (= ufl.derivative, :46)."""
    w = torch.as_tensor(np.asarray(w, dtype=np.float64), dtype=_T)
    F = ns_residual_literal(X, w, Re, **kw)
    Jm = torch.autograd.functional.jacobian(lambda ww: ns_residual_literal(X, ww, Re, **kw), w)
    return F.detach().numpy(), Jm.detach().numpy()


def stokes_matrix_literal(X):
    """16x16 bilinear form of setup_stokes_weak_form (:160-172) for one tet.

    a = inner(grad u, grad v) - inner(p, div v) + inner(div u, q) + mu_T inner(grad p, grad q),
    mu_T = 0.2 h^2, h = CellDiameter = longest vertex-vertex distance.
    """
    X = torch.as_tensor(X, dtype=_T)
    _, _, detJ, gphi = _geometry(X)
    h = max(float(torch.linalg.norm(X[a] - X[b])) for a in range(4) for b in range(a + 1, 4))
    mu_T = 0.2 * h * h                                                   # :169
    I3 = torch.eye(3, dtype=_T)
    A = torch.zeros(16, 16, dtype=_T)

    def fields(a, c, phi):
        if c < 3:
            return (phi[a] * I3[c], torch.outer(I3[c], gphi[a]), gphi[a, c],
                    torch.zeros((), dtype=_T), torch.zeros(3, dtype=_T))
        return (torch.zeros(3, dtype=_T), torch.zeros(3, 3, dtype=_T), torch.zeros((), dtype=_T),
                phi[a], gphi[a])

    for q in range(4):
        phi = _phi(torch.as_tensor(QPTS[q], dtype=_T))
        wq = QW[q] * detJ
        for a in range(4):
            for c in range(4):
                v, grad_v, div_v, qt, grad_q = fields(a, c, phi)
                for b in range(4):
                    for d in range(4):
                        u, grad_u, div_u, pt, grad_p = fields(b, d, phi)
                        t = torch.sum(grad_u * grad_v) - pt * div_v + div_u * qt \
                            + mu_T * torch.dot(grad_p, grad_q)
                        A[4 * a + c, 4 * b + d] += wq * t
    return A.numpy()
