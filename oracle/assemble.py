"""Global assembly + Dirichlet semantics of the reference, on the CPU (scipy CSR).

Test infrastructure (see oracle/__init__.py).  Follows
NavierStokes/NavierStokesChannelFlow.py:
  * .J callback :69-75   ``assemble_matrix(J, a, bcs)``: rows AND columns of
    constrained dofs zeroed, diagonal 1.
  * .F callback :51-67   assemble_vector; ``apply_lifting(F,[a],[bc],[x],-1)``
    => F += A0[:,B] (g - x_B); ``set_bc(F, bc, x, -1)`` => F_B = x_B - g.
  * LinearProblem :198-214 (Stokes): b = -A0[:,B] g on free rows, b_B = g.
Dof numbering: 4*node + c.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import element as el


def _dof_index(tets):
    """(E,16) global dof of local dof 4a+c."""
    return (4 * tets.astype(np.int64)[:, :, None] + np.arange(4)[None, None, :]).reshape(len(tets), 16)


def _coo(tets, Ae, ndof):
    dofs = _dof_index(tets)
    rows = np.repeat(dofs, 16, axis=1).ravel()
    cols = np.tile(dofs, (1, 16)).ravel()
    return sp.coo_matrix((Ae.reshape(-1), (rows, cols)), shape=(ndof, ndof)).tocsr()


def _apply_bc_matrix(A0, mask):
    free = sp.diags((1 - mask).astype(np.float64))
    return (free @ A0 @ free + sp.diags(mask.astype(np.float64))).tocsr()


def raw_ns(points, tets, w, Re, want_jac=True, chunk=200_000):
    """Unconstrained global residual (ndof,) and Jacobian (CSR or None)."""
    ndof = 4 * len(points)
    F = np.zeros(ndof)
    J = None
    W = w.reshape(-1, 4)
    for s in range(0, len(tets), chunk):
        t = tets[s:s + chunk]
        R, Je = el.ns_element(points[t], W[t], Re, want_jac=want_jac)
        np.add.at(F, _dof_index(t).ravel(), R.reshape(-1))
        if want_jac:
            Jc = _coo(t, Je.reshape(len(t), 16, 16), ndof)
            J = Jc if J is None else J + Jc
    return F, J


def assemble_ns(points, tets, w, Re, mask, g):
    """(J, F) exactly as SNES sees them after the reference's F/J callbacks."""
    F, J0 = raw_ns(points, tets, w, Re, want_jac=True)
    B = mask.astype(bool)
    F = F + J0[:, B] @ (g[B] - w[B])                # apply_lifting(..., x0=[x], alpha=-1)  :65
    F[B] = w[B] - g[B]                              # set_bc(F, bc, x, -1)                  :67
    return _apply_bc_matrix(J0, mask), F


def residual_ns(points, tets, w, Re, mask, g):
    """F only (line search); same lifting term as assemble_ns."""
    B = mask.astype(bool)
    if np.any(w[B] != g[B]):
        return assemble_ns(points, tets, w, Re, mask, g)[1]
    F, _ = raw_ns(points, tets, w, Re, want_jac=False)
    F[B] = 0.0
    return F


def assemble_stokes(points, tets, mask, g):
    """(A, b) of LinearProblem(a, L, bcs) with f = 0 (:166,171,198-214)."""
    ndof = 4 * len(points)
    A0 = None
    for s in range(0, len(tets), 200_000):
        t = tets[s:s + 200_000]
        Ac = _coo(t, el.stokes_element(points[t]).reshape(len(t), 16, 16), ndof)
        A0 = Ac if A0 is None else A0 + Ac
    B = mask.astype(bool)
    b = -(A0[:, B] @ g[B])
    b[B] = g[B]
    return _apply_bc_matrix(A0, mask), b
