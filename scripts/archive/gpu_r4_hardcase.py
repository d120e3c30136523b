"""Round 4: the case the automatic damping fails on (jittered 120x30x30 duct, Re 200; tests/test_gpu_parity.py::
test_damping_backoff_rescues_a_failed_linear_solve): per option set the hierarchy's damping, and the first Newton step's linear solve."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem

cells = tuple(int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "120,30,30").split(","))
jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
m = M.duct_mesh(cells, 4.0, jitter=jit)
bcs = B.duct_bcs(m)
sets = [("round 3", dict(amg_block_smooth=0, amg_dense_rows=0)),
        ("dense only", dict(amg_block_smooth=0)),
        ("default", dict()),
        ("growth check >=3 sweeps", dict(amg_growth_check=1)),
        ("no growth check", dict(amg_growth_check=0))]
for name, opts in sets:
    for retry in (0, 1):
        P = FlowProblem(m, bcs, reynolds=200.0, ksp_max_it=600, amg_retry_damping=retry, **opts)
        U, r = P.stokes_solve()
        F = P.zeros()
        P.jacobian(U, "ns", residual_out=F)
        P.reset_timings()
        t0 = time.perf_counter()
        y, k = P.krylov_solve(F)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = P.counters()
        h = P.hierarchy()
        print(f"{name:16s} retry {retry}: stokes {r.its:3d} its | NS step 1: {k.its:4d} its reason {k.reason} retries {c['damping_retries']} "
              f"first reason {c['first_attempt_reason']} {dt * 1e3:7.1f} ms | rows {[x['rows'] for x in h]} sweeps {[x['sweeps'] for x in h]} "
              f"omega {[round(x['omega'], 3) for x in h]}", flush=True)
        P.close()
