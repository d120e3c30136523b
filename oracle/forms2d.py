"""2-D triangle P1-P1 forms of the reference: literal, term-by-term evaluation + autograd Jacobian.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, operator by operator,
  * NS, h-based Tezduyar UGN stabilisation:
      LidDrivenFlow/LidDrivenNavierStokesFlow.py:123-143
      NavierStokes/Validation_Flow/DFG_2D_Validation.py:141-163          (identical text)
  * Stokes with pressure stabilisation mu_T (grad p, grad q):
      DFG_2D_Validation.py:107-117   (unit viscosity, mu_T = 0.2 h^2)
      LidDrivenNavierStokesFlow.py:96-109 (viscosity nu, mu_T = a0 h^2 / (4 nu), a0 = 1/3)
  * the drag / lift functional of DFG_2D_Validation.py:195-200
with explicit test functions; the Jacobian is the reverse-mode derivative of the residual with respect to
the 9 nodal coefficients of a triangle, i.e. ``ufl.derivative(a, w, dw)`` (:145 / :167).  Everything is
vectorised over triangles with ``torch.func.vmap``; no closed form is shared with the HIP kernels.

THIS is the part of the oracle the reference itself pins: DFG_2D_Validation.py:202-203 holds the
benchmark constants C_d = 5.57953523384, C_l = 0.010618948146 for exactly this form and functional.

UFL conventions (fenics-ufl 2024.2.0): grad(f)[..., j] = d f[...]/dx_j, nabla_grad = its transpose,
dot contracts last with first index, inner = full contraction, div(v) = sum_i dv_i/dx_i,
CellDiameter = longest vertex-vertex distance, conditional(le(a, b), t, f) = t if a <= b else f.
dx(degree 2) on a triangle = basix' default 3-point rule (Strang-Fix: (1/6,1/6), (1/6,2/3), (2/3,1/6),
weights 1/6) [from memory of basix 0.9 quadrature.cpp -- the libraries are absent here].

One deliberate deviation, at a measure-zero set: sqrt(dot(u,u)) has no derivative at u = 0; UFL/FFCx
produce 0/0 = NaN there, which the reference only survives because such quadrature points occur in
triangles whose three vertices all carry Dirichlet velocities, whose NaN entries ``assemble_matrix``
overwrites with zeros.  Here d|u| := 0 at u = 0.

Local dof order 3*a + c, a = cell-local vertex, c in (ux, uy, p).  Global layout (shared with the
product): 4 dofs per node [ux, uy, uz, p] with uz an identity row (Dirichlet 0).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

_T = torch.float64
QPTS = np.array([[1 / 6, 1 / 6], [1 / 6, 2 / 3], [2 / 3, 1 / 6]])
QW = np.full(3, 1.0 / 6.0)
GHAT = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])


def _phi(xi):
    return torch.stack([1.0 - xi[0] - xi[1], xi[0], xi[1]])


def _geometry(X):
    J = torch.stack([X[1] - X[0], X[2] - X[0]], dim=1)                   # J[i,j] = dx_i/dX_j
    K = torch.linalg.inv(J)
    detJ = torch.abs(torch.linalg.det(J))
    gphi = torch.as_tensor(GHAT, dtype=_T) @ K                           # d phi_a / dx_j
    e = torch.stack([X[0] - X[1], X[0] - X[2], X[1] - X[2]])
    h = torch.sqrt((e * e).sum(dim=1)).max()                             # CellDiameter
    return detJ, gphi, h


def _safe_norm(u):
    uu = torch.dot(u, u)
    pos = uu > 0
    return torch.where(pos, torch.sqrt(torch.where(pos, uu, torch.ones_like(uu))), torch.zeros_like(uu))


# Perturbations of the form for the study "what do the reference-held constants catch" (oracle/experiments/r5_pin_variants_2d.py;
# the defaults ARE the reference's form): factor on tau_LSIC, sign of the PSPG term, the coefficient 4 of tau_SUNG3 = h^2 / (4 nu),
# the 1-point centroid rule in place of the 3-point rule of dx(degree 2).
VARIANT = dict(lsic=1.0, pspg=1.0, sung3=4.0, one_point=False)


def ugn_residual_one(X, w, nu):
    """9-vector a(w; v, q) for one triangle.  X (3,2), w (9,) = [ux,uy,p]*3 (torch fp64)."""
    detJ, gphi, h = _geometry(X)
    V = VARIANT
    W = w.reshape(3, 3)
    U, P = W[:, :2], W[:, 2]
    grad_u = torch.einsum("ai,aj->ij", U, gphi)                          # grad(u)[i,j]
    nabla_grad_u = grad_u.T
    div_u = torch.trace(grad_u)
    grad_p = torch.einsum("a,aj->j", P, gphi)
    I2 = torch.eye(2, dtype=_T)
    r = 2
    total = torch.zeros(9, dtype=_T)
    for q in range(3):
        phi = _phi(torch.as_tensor([1 / 3, 1 / 3] if V["one_point"] else QPTS[q], dtype=_T))
        u = torch.einsum("a,ai->i", phi, U)
        p = torch.dot(phi, P)
        u_norm = _safe_norm(u)                                           # sqrt(dot(u,u))
        safe_un = torch.where(u_norm > 1e-8, u_norm, torch.ones_like(u_norm))
        tau_SUNG1 = h / (2 * safe_un)
        inv_tau_SUNG1 = torch.where(u_norm <= 1e-8, torch.zeros_like(u_norm), 1 / (tau_SUNG1 ** r))
        tau_SUNG3 = h * h / (V["sung3"] * nu)
        tau_SUPG = (inv_tau_SUNG1 + 1 / (tau_SUNG3 ** r)) ** (-1 / r)
        Re_UGN = u_norm * h / (2 * nu)
        z = torch.where(Re_UGN <= 3, Re_UGN / 3, torch.ones_like(Re_UGN))
        tau_LSIC = V["lsic"] * h / 2 * u_norm * z
        conv = u @ nabla_grad_u                                          # dot(u, nabla_grad(u))
        # P1 on an affine cell: div(sym(grad(u))) = 0
        res = conv + grad_p
        wq = QW[q] * detJ
        vals = []
        for a in range(3):
            for c in range(3):
                if c < 2:
                    v = phi[a] * I2[c]
                    grad_v = torch.outer(I2[c], gphi[a])
                    div_v = gphi[a, c]
                    qt = torch.zeros((), dtype=_T)
                    grad_q = torch.zeros(2, dtype=_T)
                else:
                    v = torch.zeros(2, dtype=_T)
                    grad_v = torch.zeros(2, 2, dtype=_T)
                    div_v = torch.zeros((), dtype=_T)
                    qt = phi[a]
                    grad_q = gphi[a]
                t = torch.dot(conv, v)                                   # advection
                t = t + nu * torch.sum(grad_u * grad_v)                  # diffusion
                t = t - p * div_v                                        # pressure
                t = t + qt * div_u                                       # incompressibility
                t = t + tau_SUPG * torch.dot(u @ grad_v.T, res)          # SUPG: dot(u, nabla_grad(v))
                t = t + V["pspg"] * tau_SUPG * torch.dot(grad_q, res)    # PSPG
                t = t + tau_LSIC * div_v * div_u                         # LSIC
                vals.append(wq * t)
        total = total + torch.stack(vals)
    return total


def stokes_matrix_one(X, nu_s, beta):
    """9x9 Stokes bilinear form: nu_s (grad u, grad v) - (p, div v) + (div u, q) + beta h^2 (grad p, grad q)."""
    detJ, gphi, h = _geometry(X)
    mu_T = beta * h * h

    def lin(w):
        W = w.reshape(3, 3)
        U, P = W[:, :2], W[:, 2]
        grad_u = torch.einsum("ai,aj->ij", U, gphi)
        div_u = torch.trace(grad_u)
        grad_p = torch.einsum("a,aj->j", P, gphi)
        I2 = torch.eye(2, dtype=_T)
        total = torch.zeros(9, dtype=_T)
        for q in range(3):
            phi = _phi(torch.as_tensor(QPTS[q], dtype=_T))
            p = torch.dot(phi, P)
            wq = QW[q] * detJ
            vals = []
            for a in range(3):
                for c in range(3):
                    if c < 2:
                        t = nu_s * torch.sum(grad_u * torch.outer(I2[c], gphi[a])) - p * gphi[a, c]
                    else:
                        t = phi[a] * div_u + mu_T * torch.dot(grad_p, gphi[a])
                    vals.append(wq * t)
            total = total + torch.stack(vals)
        return total

    return torch.func.jacrev(lin)(torch.zeros(9, dtype=_T))


def ugn_elements(points, tris, w4, nu, want_jac=True, chunk=20000):
    """Element residuals (E,9) and Jacobians (E,9,9) of every triangle; w4 = global [ux,uy,uz,p] vector."""
    W4 = np.asarray(w4, dtype=np.float64).reshape(-1, 4)
    R, Jm = [], []
    f = torch.func.vmap(lambda X, w: ugn_residual_one(X, w, nu))
    fj = torch.func.vmap(torch.func.jacrev(lambda X, w: ugn_residual_one(X, w, nu), argnums=1))
    for s in range(0, len(tris), chunk):
        t = tris[s:s + chunk]
        X = torch.as_tensor(points[t][:, :, :2], dtype=_T)
        w = torch.as_tensor(W4[t][:, :, [0, 1, 3]].reshape(len(t), 9), dtype=_T)
        R.append(f(X, w).numpy())
        if want_jac:
            Jm.append(fj(X, w).numpy())
    return np.concatenate(R), (np.concatenate(Jm) if want_jac else None)


def stokes_elements(points, tris, nu_s, beta, chunk=20000):
    out = []
    f = torch.func.vmap(lambda X: stokes_matrix_one(X, nu_s, beta))
    for s in range(0, len(tris), chunk):
        X = torch.as_tensor(points[tris[s:s + chunk]][:, :, :2], dtype=_T)
        out.append(f(X).numpy())
    return np.concatenate(out)


# ---- global assembly in the product's 4-dof-per-node layout (uz = identity row) -------------------------
_C4 = np.array([0, 1, 3])


def _dofs(tris):
    return (4 * tris.astype(np.int64)[:, :, None] + _C4[None, None, :]).reshape(len(tris), 9)


def _coo(tris, Ae, ndof):
    d = _dofs(tris)
    rows = np.repeat(d, 9, axis=1).ravel()
    cols = np.tile(d, (1, 9)).ravel()
    return sp.coo_matrix((Ae.reshape(-1), (rows, cols)), shape=(ndof, ndof)).tocsr()


def full_mask(mask):
    """The product forces the unused z component to a homogeneous Dirichlet dof."""
    m = np.array(mask, dtype=np.uint8).copy()
    m[2::4] = 1
    return m


def _apply_bc_matrix(A0, mask):
    free = sp.diags((1 - mask).astype(np.float64))
    return (free @ A0 @ free + sp.diags(mask.astype(np.float64))).tocsr()


def assemble_ugn(points, tris, w, nu, mask, g):
    """(J, F) as the reference's NonlinearProblem hands them to the Newton solver: rows and columns of
    constrained dofs zeroed with unit diagonal, F += A0[:,B](g - x_B) (lifting), F_B = x_B - g."""
    mask = full_mask(mask)
    g = np.where(np.arange(len(g)) % 4 == 2, 0.0, g)
    ndof = 4 * len(points)
    R, Je = ugn_elements(points, tris, w, nu)
    F = np.zeros(ndof)
    np.add.at(F, _dofs(tris).ravel(), R.reshape(-1))
    J0 = _coo(tris, Je, ndof)
    B = mask.astype(bool)
    F = F + J0[:, B] @ (g[B] - w[B])
    F[B] = w[B] - g[B]
    return _apply_bc_matrix(J0, mask), F


def residual_ugn(points, tris, w, nu, mask, g):
    mask = full_mask(mask)
    g = np.where(np.arange(len(g)) % 4 == 2, 0.0, g)
    B = mask.astype(bool)
    if np.any(w[B] != g[B]):
        return assemble_ugn(points, tris, w, nu, mask, g)[1]
    R, _ = ugn_elements(points, tris, w, nu, want_jac=False)
    F = np.zeros(4 * len(points))
    np.add.at(F, _dofs(tris).ravel(), R.reshape(-1))
    F[B] = 0.0
    return F


def assemble_stokes2d(points, tris, mask, g, nu_s=1.0, beta=0.2):
    """(A, b) of LinearProblem(a, L, bcs) with f = 0."""
    mask = full_mask(mask)
    g = np.where(np.arange(len(g)) % 4 == 2, 0.0, g)
    ndof = 4 * len(points)
    A0 = _coo(tris, stokes_elements(points, tris, nu_s, beta), ndof)
    B = mask.astype(bool)
    b = -(A0[:, B] @ g[B])
    b[B] = g[B]
    return _apply_bc_matrix(A0, mask), b


def solve_stokes2d(points, tris, mask, g, nu_s=1.0, beta=0.2):
    import scipy.sparse.linalg as spla
    A, b = assemble_stokes2d(points, tris, mask, g, nu_s, beta)
    return spla.splu(sp.csc_matrix(A)).solve(b)


def newton2d(points, tris, w0, nu, mask, g, *, rtol=1e-9, atol=1e-10, max_it=50, monitor=None):
    """dolfinx NewtonSolver as the reference configures it (LidDrivenNavierStokesFlow.py:154-169,
    DFG_2D_Validation.py:171-187): full steps, convergence_criterion 'incremental' (||dx|| < atol or
    ||dx|| / ||dx_0|| < rtol), linear solves by sparse LU.  Returns (w, info)."""
    import scipy.sparse.linalg as spla
    w = w0.copy()
    dx0 = None
    hist = []
    for it in range(1, max_it + 1):
        J, F = assemble_ugn(points, tris, w, nu, mask, g)
        dx = spla.splu(sp.csc_matrix(J)).solve(F)
        w = w - dx
        r = float(np.linalg.norm(dx))
        hist.append((float(np.linalg.norm(F)), r))
        if monitor:
            monitor(it, hist[-1])
        if dx0 is None:
            dx0 = r
        if r < atol or r / dx0 < rtol:
            return w, dict(its=it, converged=True, hist=hist)
    return w, dict(its=max_it, converged=False, hist=hist)


# ---- drag / lift of DFG_2D_Validation.py:195-200 by plain loops over the obstacle edges ------------------
def drag_lift_loops(points, tris, edges, w, nu, U_mean=0.2, L=0.1):
    """n = -FacetNormal (pointing out of the obstacle into the fluid), u_t = (n_y, -n_x).u,
    C_D =  2/(U^2 L) int nu (grad(u_t).n) n_y - p n_x ds,   C_L = -2/(U^2 L) int nu (grad(u_t).n) n_x + p n_y ds
    (the script writes 2 / (0.1 * 0.2**2)); 2-point Gauss rule on every edge."""
    W = np.asarray(w, dtype=np.float64).reshape(-1, 4)
    tsets = [set(int(v) for v in t) for t in tris]
    gq = [0.5 - 0.5 / np.sqrt(3.0), 0.5 + 0.5 / np.sqrt(3.0)]
    cd = cl = 0.0
    for e in edges:
        a, b = int(e[0]), int(e[1])
        par = [k for k, s in enumerate(tsets) if a in s and b in s]
        assert len(par) == 1, "obstacle edge must belong to exactly one triangle"
        t = [int(v) for v in tris[par[0]]]
        X = points[t][:, :2]
        A = np.hstack([np.ones((3, 1)), X])
        grads = np.linalg.inv(A)[1:, :].T                               # (3,2) d phi_a / dx
        gu = sum(np.outer(W[t[k], :2], grads[k]) for k in range(3))       # grad(u)[i,j]
        pa, pb = points[a][:2], points[b][:2]
        tv = pb - pa
        ln = float(np.linalg.norm(tv))
        nf = np.array([tv[1], -tv[0]]) / ln
        opp = [v for v in t if v not in (a, b)][0]
        if np.dot(nf, pa - points[opp][:2]) < 0:
            nf = -nf                                                     # FacetNormal: out of the fluid domain
        n = -nf
        tt = np.array([n[1], -n[0]])
        dut_dn = float(tt @ gu @ n)                                      # inner(grad(u_t), n)
        for s in gq:
            p = (1 - s) * W[a, 3] + s * W[b, 3]
            cd += 0.5 * ln * (nu * dut_dn * n[1] - p * n[0])
            cl += 0.5 * ln * (nu * dut_dn * n[0] + p * n[1])
    c = 2.0 / (U_mean ** 2 * L)
    return c * cd, -c * cl
