"""ctypes wrapper of oracle/c/liboracle.so (the C/OpenMP restatement).  Test infrastructure."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "c", "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "c", "sns_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")])
        _lib = C.CDLL(LIB)
        _lib.orc_pattern.restype = C.c_int64
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def num_threads() -> int:
    return load().orc_num_threads()


def set_num_threads(n: int):
    load().orc_set_num_threads(C.c_int(int(n)))


def ns_elements(X, W, Re):
    lib = load()
    X = np.ascontiguousarray(X, np.float64); W = np.ascontiguousarray(W, np.float64).reshape(len(X), 16)
    R = np.empty((len(X), 16)); J = np.empty((len(X), 16, 16))
    lib.orc_ns_elements(C.c_int64(len(X)), _p(X), _p(W), C.c_double(Re), _p(R), _p(J))
    return R, J


def stokes_elements(X):
    lib = load()
    X = np.ascontiguousarray(X, np.float64)
    A = np.empty((len(X), 16, 16))
    lib.orc_stokes_elements(C.c_int64(len(X)), _p(X), _p(A))
    return A


def pattern(n, tets):
    lib = load()
    tets = np.ascontiguousarray(tets, np.int32)
    nnzb = lib.orc_pattern(C.c_int32(n), C.c_int64(len(tets)), _p(tets), None, None)
    rowptr = np.empty(n + 1, np.int32); colind = np.empty(nnzb, np.int32)
    lib.orc_pattern(C.c_int32(n), C.c_int64(len(tets)), _p(tets), _p(rowptr), _p(colind))
    return rowptr, colind


def assemble(form, pts, tets, w, Re, mask, g, rowptr, colind):
    lib = load()
    pts = np.ascontiguousarray(pts, np.float64); tets = np.ascontiguousarray(tets, np.int32)
    mask = np.ascontiguousarray(mask, np.uint8); g = np.ascontiguousarray(g, np.float64)
    w = None if w is None else np.ascontiguousarray(w, np.float64)
    n = len(pts)
    vals = np.empty((len(colind), 16)); F = np.empty(4 * n)
    lib.orc_assemble(C.c_int(1 if form == "ns" else 0), C.c_int32(n), C.c_int64(len(tets)), _p(pts), _p(tets), _p(w),
                     C.c_double(Re), _p(mask), _p(g), _p(rowptr), _p(colind), _p(vals), _p(F))
    return vals, F


def to_scipy(n, rowptr, colind, vals):
    import scipy.sparse as sp
    return sp.bsr_matrix((vals.reshape(-1, 4, 4), colind, rowptr), shape=(4 * n, 4 * n)).tocsr()


def solve(n, rowptr, colind, vals, b, x0=None, method="tfqmr", pc="ilu0", nblocks=None, rtol=1e-8, atol=1e-50,
          maxit=10000):
    lib = load()
    x = np.zeros(4 * n) if x0 is None else np.array(x0, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    its, reason, rn = C.c_int(), C.c_int(), C.c_double()
    lib.orc_solve(C.c_int32(n), _p(rowptr), _p(colind), _p(np.ascontiguousarray(vals)), _p(b), _p(x),
                  C.c_int({"bicgstab": 0, "tfqmr": 1}[method]), C.c_int({"none": 0, "bjacobi": 1, "ilu0": 2}[pc]),
                  C.c_int(nblocks or num_threads()), C.c_double(rtol), C.c_double(atol), C.c_int(maxit),
                  C.byref(its), C.byref(reason), C.byref(rn))
    return x, its.value, reason.value, rn.value


def last_solve_info():
    """What the last `solve` saw at its exit: dict(tested = the value its stopping test ran on (tfqmr: PETSc's quasi-residual
    bound tau*sqrt(m+1)), petsc_criterion_met, true_residual = ||b - A x||, bnorm)."""
    lib = load()
    out = (C.c_double * 4)()
    lib.orc_last_solve_info(out)
    return {"tested": out[0], "petsc_criterion_met": bool(out[1]), "true_residual": out[2], "bnorm": out[3]}
