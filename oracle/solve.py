"""Linear and nonlinear solves of the reference path, on the CPU.

Test infrastructure (see oracle/__init__.py).
  * solve_stokes    LinearProblem of NavierStokesChannelFlow.py:197-218
  * newton          PETSc.SNES newtonls + bt line search as configured at
                    :274-283 (rtol=atol=1e-8, stol 1e-8, max_it 30; KSP rtol 1e-8)
  * bicgstab_bj     point-block-Jacobi (4x4 nodal blocks) preconditioned
                    BiCGStab: the algorithm the HIP Krylov driver implements,
                    so iteration histories can be compared, not only fields.
The tightly converged reference fields come from sparse LU (splu).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import assemble as asm

# PETSc converged-reason codes mirrored by the product (sns.h)
KSP_CONVERGED_RTOL, KSP_CONVERGED_ATOL = 2, 3
KSP_DIVERGED_ITS, KSP_DIVERGED_BREAKDOWN = -3, -5
SNES_CONVERGED_FNORM_ABS, SNES_CONVERGED_FNORM_RELATIVE, SNES_CONVERGED_SNORM_RELATIVE = 2, 3, 4
SNES_DIVERGED_LINEAR_SOLVE, SNES_DIVERGED_MAX_IT, SNES_DIVERGED_LINE_SEARCH = -3, -5, -6


def lu_solve(A, b):
    return spla.splu(sp.csc_matrix(A)).solve(b)


def block_jacobi_inverse(A):
    """(n,4,4) inverses of the 4x4 nodal diagonal blocks of CSR A."""
    n = A.shape[0] // 4
    Ab = sp.bsr_matrix(A, blocksize=(4, 4))
    Ab.sort_indices()
    D = np.zeros((n, 4, 4))
    for i in range(n):
        s, e = Ab.indptr[i], Ab.indptr[i + 1]
        k = np.searchsorted(Ab.indices[s:e], i)
        D[i] = Ab.data[s + k]
    return np.linalg.inv(D)


def bicgstab_bj(A, b, x0=None, rtol=1e-8, atol=1e-50, maxit=10000, Dinv=None, history=None):
    """Right-preconditioned BiCGStab, M = blockdiag_4x4(A).  Returns (x, its, reason).

    Stopping: ||r|| <= max(rtol*||b||, atol) on the TRUE residual recurrence.
    """
    n = A.shape[0]
    if Dinv is None:
        Dinv = block_jacobi_inverse(A)

    def M(v):
        return np.einsum("nij,nj->ni", Dinv, v.reshape(-1, 4)).reshape(n)

    x = np.zeros(n) if x0 is None else x0.copy()
    r = b - A @ x
    bnorm = np.linalg.norm(b)
    tol = max(rtol * bnorm, atol)
    rn = np.linalg.norm(r)
    if history is not None:
        history.append(rn)
    if rn <= tol:
        return x, 0, (KSP_CONVERGED_ATOL if rn <= atol else KSP_CONVERGED_RTOL)
    rhat = r.copy()
    rho = alpha = omega = 1.0
    v = np.zeros(n)
    p = np.zeros(n)
    for it in range(1, maxit + 1):
        rho_new = rhat @ r
        if rho_new == 0.0:
            return x, it, KSP_DIVERGED_BREAKDOWN
        beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v)
        ph = M(p)
        v = A @ ph
        alpha = rho_new / (rhat @ v)
        s = r - alpha * v
        sh = M(s)
        t = A @ sh
        tt = t @ t
        omega = (t @ s) / tt if tt > 0 else 0.0
        x = x + alpha * ph + omega * sh
        r = s - omega * t
        rho = rho_new
        rn = np.linalg.norm(r)
        if history is not None:
            history.append(rn)
        if rn <= tol:
            return x, it, (KSP_CONVERGED_ATOL if rn <= atol else KSP_CONVERGED_RTOL)
        if omega == 0.0:
            return x, it, KSP_DIVERGED_BREAKDOWN
    return x, maxit, KSP_DIVERGED_ITS


def solve_stokes(points, tets, mask, g, method="lu", **kw):
    """U of solve_stokes_problem (:197-218).  method 'lu' = tight reference,
    'bicgstab' = the product's Krylov settings."""
    A, b = asm.assemble_stokes(points, tets, mask, g)
    if method == "lu":
        return lu_solve(A, b), {"its": 0, "reason": KSP_CONVERGED_RTOL}
    x, its, reason = bicgstab_bj(A, b, **kw)
    return x, {"its": its, "reason": reason}


def newton(points, tets, w0, Re, mask, g, *, rtol=1e-8, atol=1e-8, stol=1e-8, max_it=30,
           linear="lu", ksp_rtol=1e-8, ksp_maxit=10000, ls_alpha=1e-4, ls_max_it=40, monitor=None):
    """SNES newtonls + bt as set up at :274-283.  Returns (w, info).

    info: its, reason, fnorms (per iteration, incl. initial), ksp_its, lambdas.
    """
    w = w0.copy()
    fnorms, ksp_its, lambdas = [], [], []
    J, F = asm.assemble_ns(points, tets, w, Re, mask, g)
    f = np.linalg.norm(F)
    f0 = f
    fnorms.append(f)
    if monitor:
        monitor(0, f)
    if f < atol:
        return w, dict(its=0, reason=SNES_CONVERGED_FNORM_ABS, fnorms=fnorms, ksp_its=ksp_its, lambdas=lambdas)
    for it in range(1, max_it + 1):
        if linear == "lu":
            y = lu_solve(J, F)
            ksp_its.append(0)
        else:
            y, k, reason = bicgstab_bj(J, F, rtol=ksp_rtol, maxit=ksp_maxit)
            ksp_its.append(k)
            if reason < 0:
                return w, dict(its=it, reason=SNES_DIVERGED_LINEAR_SOLVE, fnorms=fnorms, ksp_its=ksp_its,
                               lambdas=lambdas)
        # ---- bt line search (cubic), x_new = x - lambda y ----
        initslope = F @ (J @ y)
        if initslope > 0:
            initslope = -initslope
        if initslope == 0:
            initslope = -1.0
        lam = 1.0
        wn = w - lam * y
        Fn = asm.residual_ns(points, tets, wn, Re, mask, g)
        gn = np.linalg.norm(Fn)
        ok = 0.5 * gn * gn <= 0.5 * f * f + lam * ls_alpha * initslope
        if not ok:
            lamprev, gprev = lam, gn
            lamtemp = -initslope / (gn * gn - f * f - 2.0 * lam * initslope)
            lam = min(max(lamtemp, 0.1 * lam), 0.5 * lam)
            for _ in range(ls_max_it):
                wn = w - lam * y
                Fn = asm.residual_ns(points, tets, wn, Re, mask, g)
                gn = np.linalg.norm(Fn)
                if 0.5 * gn * gn <= 0.5 * f * f + lam * ls_alpha * initslope:
                    ok = True
                    break
                t1 = 0.5 * (gn * gn - f * f) - lam * initslope
                t2 = 0.5 * (gprev * gprev - f * f) - lamprev * initslope
                a = (t1 / lam ** 2 - t2 / lamprev ** 2) / (lam - lamprev)
                bq = (-lamprev * t1 / lam ** 2 + lam * t2 / lamprev ** 2) / (lam - lamprev)
                d = max(bq * bq - 3 * a * initslope, 0.0)
                lamtemp = -initslope / (2.0 * bq) if a == 0 else (-bq + np.sqrt(d)) / (3.0 * a)
                lamprev, gprev = lam, gn
                lam = min(max(lamtemp, 0.1 * lam), 0.5 * lam)
            if not ok:
                return w, dict(its=it, reason=SNES_DIVERGED_LINE_SEARCH, fnorms=fnorms, ksp_its=ksp_its,
                               lambdas=lambdas)
        lambdas.append(lam)
        snorm = lam * np.linalg.norm(y)
        w = wn
        xnorm = np.linalg.norm(w)
        f = gn
        fnorms.append(f)
        if monitor:
            monitor(it, f)
        if f < atol:
            reason = SNES_CONVERGED_FNORM_ABS
        elif f <= rtol * f0:
            reason = SNES_CONVERGED_FNORM_RELATIVE
        elif snorm < stol * xnorm:
            reason = SNES_CONVERGED_SNORM_RELATIVE
        else:
            reason = 0
        if reason:
            return w, dict(its=it, reason=reason, fnorms=fnorms, ksp_its=ksp_its, lambdas=lambdas)
        J, F = asm.assemble_ns(points, tets, w, Re, mask, g)
    return w, dict(its=max_it, reason=SNES_DIVERGED_MAX_IT, fnorms=fnorms, ksp_its=ksp_its, lambdas=lambdas)
