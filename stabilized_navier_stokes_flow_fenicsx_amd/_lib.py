"""ctypes binding of the C-ABI in include/sns.h (libsns.so, built in-tree by
``__graft_entry__.build()`` / ``csrc/Makefile``).

There is NO fallback: if the HIP library is missing or fails to load, importing
the solver raises.  torch is imported first so that libsns.so binds to the same
HIP / RCCL runtime instances torch already loaded (same SONAMEs).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsns.so")


class SnsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"sns error {code}: {msg}")
        self.code = code


class SnsOptions(C.Structure):
    """Mirror of ``sns_options`` (include/sns.h)."""
    _fields_ = [
        ("reynolds", C.c_double), ("ksp_type", C.c_int), ("pc_type", C.c_int),
        ("ksp_rtol", C.c_double), ("ksp_atol", C.c_double), ("ksp_max_it", C.c_int),
        ("gmres_restart", C.c_int), ("snes_rtol", C.c_double), ("snes_atol", C.c_double),
        ("snes_stol", C.c_double), ("snes_max_it", C.c_int), ("amg_max_levels", C.c_int),
        ("amg_coarse_size", C.c_int), ("amg_agg_size", C.c_int), ("amg_nu", C.c_int),
        ("amg_omega", C.c_double), ("monitor", C.c_int), ("corrected_convection", C.c_int), ("amg_f32_matrix", C.c_int), ("amg_nu_coarse", C.c_int), ("amg_nu_deep", C.c_int), ("amg_nu_l2", C.c_int), ("amg_sweep_exchange_rows", C.c_int), ("amg_replicate_rows", C.c_int), ("amg_post_exchange", C.c_int), ("assembly_fused", C.c_int),
        ("stokes_viscosity", C.c_double), ("stokes_beta", C.c_double),
        ("amg_nu_l1_pre", C.c_int),
        ("amg_nu_l1_post", C.c_int),
        ("amg_retry_damping", C.c_int),
        ("amg_retry_stall_its", C.c_int),
        ("halo_overlap", C.c_int),
        ("amg_fused_post", C.c_int),
        ("amg_nu_scale_with_size", C.c_int),
        ("amg_dense_rows", C.c_int),
        ("amg_block_smooth", C.c_int),
        ("amg_bnu_l1", C.c_int),
        ("amg_bnu_l2", C.c_int),
        ("amg_bnu_deep", C.c_int),
        ("amg_ritz_limit", C.c_int),
        ("amg_block_max_rows", C.c_int),
        ("amg_block_fine_rows", C.c_int),
        ("amg_fuse_restrict", C.c_int),
        ("halo_windows", C.c_int),
        ("amg_exact_sweeps", C.c_int),
    ]


class SnsTimings(C.Structure):
    _fields_ = [("assemble_ms", C.c_double), ("pc_setup_ms", C.c_double), ("krylov_ms", C.c_double),
                ("spmv_ms_avg", C.c_double), ("spmv_calls", C.c_int64), ("ksp_its", C.c_int),
                ("amg_levels", C.c_int)]


# constants of sns.h
ABI_VERSION = 7                          # SNS_ABI_VERSION of the header this mirror was written against
FORM_STOKES, FORM_NS = 0, 1
KSP_BICGSTAB, KSP_FGMRES, KSP_TFQMR = 0, 1, 2
PC_NONE, PC_BJACOBI, PC_AMG = 0, 1, 2
EXPORT_ROWPTR, EXPORT_COLIND, EXPORT_VALS, EXPORT_KE, EXPORT_FE = 0, 1, 2, 3, 4
KSP_NAMES = {"bicgstab": KSP_BICGSTAB, "bcgs": KSP_BICGSTAB, "fgmres": KSP_FGMRES, "gmres": KSP_FGMRES, "tfqmr": KSP_TFQMR}
PC_NAMES = {"none": PC_NONE, "bjacobi": PC_BJACOBI, "jacobi": PC_BJACOBI, "amg": PC_AMG}

# every symbol include/sns.h declares: (name, restype, argtypes)
_H = C.c_void_p
_P = C.c_void_p
_SIGNATURES = [
    ("sns_default_options", None, [C.POINTER(SnsOptions)]),
    ("sns_last_error", C.c_char_p, []),
    ("sns_version", C.c_char_p, []),
    ("sns_abi_version", C.c_int, []),
    ("sns_options_size", C.c_int64, []),
    ("sns_create", C.c_int, [C.POINTER(_H), C.c_int32, C.c_int64, _P, _P, _P, _P, C.c_int, C.POINTER(SnsOptions)]),
    ("sns_create_2d", C.c_int, [C.POINTER(_H), C.c_int32, C.c_int64, _P, _P, _P, _P, C.c_int, C.POINTER(SnsOptions)]),
    ("sns_destroy", C.c_int, [_H]),
    ("sns_set_stream", C.c_int, [_H, _P]),
    ("sns_set_options", C.c_int, [_H, C.POINTER(SnsOptions)]),
    ("sns_get_options", C.c_int, [_H, C.POINTER(SnsOptions)]),
    ("sns_set_form_variant", C.c_int, [_H, C.c_double, C.c_double, C.c_double, C.c_int]),
    ("sns_get_sizes", C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int64)]),
    ("sns_comm_unique_id", C.c_int, [C.c_char_p]),
    ("sns_attach_comm", C.c_int, [_H, C.c_int, C.c_int, C.c_char_p, C.c_int32, C.c_int, _P, _P, _P, _P, _P]),
    ("sns_team_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("sns_team_destroy", C.c_int, [_P]),
    ("sns_attach_team", C.c_int, [_H, _P, C.c_int, C.c_int, C.c_int32, C.c_int, _P, _P, _P, _P, _P]),
    ("sns_peer_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(_P), C.c_char_p]),
    ("sns_peer_connect", C.c_int, [_P, C.c_char_p]),
    ("sns_peer_disconnect", C.c_int, [_P]),
    ("sns_peer_destroy", C.c_int, [_P]),
    ("sns_peer_check_links", C.c_int, [_P, C.c_int]),
    ("sns_peer_selftest", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("sns_attach_peer", C.c_int, [_H, _P, C.c_int32, C.c_int, _P, _P, _P, _P, _P]),
    ("sns_residual", C.c_int, [_H, C.c_int, _P, _P]),
    ("sns_jacobian", C.c_int, [_H, C.c_int, _P, _P]),
    ("sns_spmv", C.c_int, [_H, _P, _P]),
    ("sns_pc_setup", C.c_int, [_H]),
    ("sns_pc_apply", C.c_int, [_H, _P, _P]),
    ("sns_krylov_solve", C.c_int, [_H, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("sns_stokes_solve", C.c_int, [_H, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("sns_newton_solve", C.c_int, [_H, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_double), C.c_int]),
    ("sns_get_bsr", C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(_P), C.POINTER(_P),
                              C.POINTER(_P)]),
    ("sns_get_element_scratch", C.c_int, [_H, C.POINTER(_P), C.POINTER(_P)]),
    ("sns_export", C.c_int, [_H, C.c_int, _P, C.c_int64]),
    ("sns_get_timings", C.c_int, [_H, C.POINTER(SnsTimings)]),
    ("sns_get_counters", C.c_int, [_H, C.POINTER(C.c_int64)]),
    ("sns_comm_info", C.c_int, [_H, C.POINTER(C.c_int32)]),
    ("sns_get_hierarchy", C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    ("sns_dense_inverse", C.c_int, [C.c_int, C.c_int32, _P, _P]),
    ("sns_get_cycle", C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("sns_reset_timings", C.c_int, [_H]),
    ("sns_time_kernels", C.c_int, [_H, C.c_int]),
    ("sns_get_kernel_times", C.c_int, [_H, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("sns_bench_spmv", C.c_int, [_H, _P, _P, C.c_int, C.POINTER(C.c_double)]),
    ("sns_bench_assemble", C.c_int, [_H, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_double)]),
    ("sns_bench_collective", C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("sns_streamtrace", C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.c_double, _P, _P, _P, _P, _P]),
    ("sns_host_pattern", C.c_int, [C.c_int32, C.c_int64, _P, C.POINTER(C.c_int64), _P, _P, _P, _P]),
    ("sns_host_aggregate", C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int, _P, C.POINTER(C.c_int32)]),
    ("sns_host_aggregate_pts", C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int, _P, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("sns_host_boundary_rows", C.c_int, [C.c_int32, _P, _P, _P, C.POINTER(C.c_int32)]),
    ("sns_host_cycle_policy", C.c_int, [C.POINTER(SnsOptions), C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int64, _P, _P, _P, _P, _P]),
    ("sns_host_hessenberg_eigs", C.c_int, [C.c_int, _P, _P, _P]),
]
SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def load():
    """dlopen libsns.so and type every entry point; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                          " (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # ABI guard: sns_options grows at its end and the getters' out-arrays have grown between rounds; a mirror of another header
    # would hand the library short buffers, so refuse before any other call
    if lib.sns_abi_version() != ABI_VERSION or lib.sns_options_size() != C.sizeof(SnsOptions):
        raise ImportError(f"{LIB_PATH}: ABI {lib.sns_abi_version()} / sizeof(sns_options) {lib.sns_options_size()} does not match "
                          f"this binding ({ABI_VERSION} / {C.sizeof(SnsOptions)}): rebuild the library or update _lib.py")
    if hasattr(lib, "sns_bench_variants"):           # experiment build only (make HARNESS=1, csrc/sns_harness.h)
        lib.sns_bench_variants.restype = C.c_int
        lib.sns_bench_variants.argtypes = [_H, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise SnsError(rc, load().sns_last_error().decode())


def default_options(**kw) -> SnsOptions:
    o = SnsOptions()
    load().sns_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "ksp_type" and isinstance(v, str):
            v = KSP_NAMES[v]
        if k == "pc_type" and isinstance(v, str):
            v = PC_NAMES[v]
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k}")
        setattr(o, k, v)
    return o


def host_pattern(n_nodes: int, tets):
    """(rowptr, colind, c_ptr, c_idx) of the BSR pattern + gather lists (host only)."""
    import numpy as np
    lib = load()
    tets = np.ascontiguousarray(tets, dtype=np.int32)
    nnzb = C.c_int64()
    check(lib.sns_host_pattern(n_nodes, len(tets), tets.ctypes.data, C.byref(nnzb), None, None, None, None))
    rowptr = np.empty(n_nodes + 1, np.int32)
    colind = np.empty(nnzb.value, np.int32)
    c_ptr = np.empty(nnzb.value + 1, np.int64)
    c_idx = np.empty(16 * len(tets), np.int32)
    check(lib.sns_host_pattern(n_nodes, len(tets), tets.ctypes.data, C.byref(nnzb), rowptr.ctypes.data,
                               colind.ctypes.data, c_ptr.ctypes.data, c_idx.ctypes.data))
    return rowptr, colind, c_ptr, c_idx


def host_aggregate(rowptr, colind, n_active=None, max_agg=8, pts=None):
    """(agg, n_agg) of the greedy sweep; with node coordinates ``pts`` (n, 3): (agg, n_agg, which) of the level's aggregation as
    the hierarchy build runs it (which = 0 greedy sweep, 1 pairwise: sns_host_aggregate_pts)."""
    import numpy as np
    lib = load()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    n = len(rowptr) - 1
    agg = np.empty(n, np.int32)
    nc = C.c_int32()
    if pts is None:
        check(lib.sns_host_aggregate(n, rowptr.ctypes.data, colind.ctypes.data, n if n_active is None else n_active,
                                     max_agg, agg.ctypes.data, C.byref(nc)))
        return agg, nc.value
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(n, 3)
    which = C.c_int32()
    check(lib.sns_host_aggregate_pts(n, rowptr.ctypes.data, colind.ctypes.data, n if n_active is None else n_active,
                                     max_agg, pts.ctypes.data, agg.ctypes.data, C.byref(nc), C.byref(which)))
    return agg, nc.value, which.value


def host_boundary_rows(n_owned: int, rowptr, colind):
    """Owned rows with a ghost column (interior / boundary split of the multi-GPU SpMV)."""
    import numpy as np
    lib = load()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    out = np.empty(max(1, n_owned), np.int32)
    n = C.c_int32()
    check(lib.sns_host_boundary_rows(n_owned, rowptr.ctypes.data, colind.ctypes.data, out.ctypes.data, C.byref(n)))
    return out[:n.value].copy()


def host_cycle_policy(rows_global, nranks=1, windows=False, rep_level=0, rows_global_l1=None, has_blocks=None, **opt_kw):
    """The hierarchy's policy table (sns_host_cycle_policy, csrc/sns_policy.h) for a hierarchy of the given shape: a list of
    dict(kind, pre, post, exact) per level (kind -1: the source of the replicated copy, not cycled).  No GPU needed."""
    import numpy as np
    lib = load()
    o = default_options(**opt_kw)
    rows = np.ascontiguousarray(rows_global, dtype=np.int64)
    n = len(rows)
    hb = np.ones(n, np.uint8) if has_blocks is None else np.ascontiguousarray(has_blocks, dtype=np.uint8)
    kind, pre, post, ex = (np.zeros(n, np.int32) for _ in range(4))
    l1 = int(rows[1]) if rows_global_l1 is None and n > 1 else int(rows_global_l1 or 0)
    check(lib.sns_host_cycle_policy(C.byref(o), int(nranks), int(bool(windows)), n, rows.ctypes.data, int(rep_level), l1,
                                    hb.ctypes.data, kind.ctypes.data, pre.ctypes.data, post.ctypes.data, ex.ctypes.data))
    return [dict(kind=int(kind[l]), pre=int(pre[l]), post=int(post[l]), exact=int(ex[l])) for l in range(n)]
