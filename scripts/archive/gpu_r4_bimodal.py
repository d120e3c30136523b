"""Round 4: the headline's run-to-run bimodality (137.5 / 142.5 ms per Newton iteration with the same binary on the same box): does the
mode change between problem instances of ONE process (then it follows the allocations) or only between processes?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.duct_mesh((300, 75, 75), 4.0)
bcs = B.duct_bcs(m).flatten()
keep = []
for trial in range(5):
    P = FlowProblem(m, bcs, reynolds=200.0, snes_max_it=1)
    U, r = P.stokes_solve()
    w = U.clone()
    for _ in range(2):
        w, n = P.newton_solve(w)
    P.reset_timings(); P.time_kernels(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    its = []
    w = U.clone()
    for _ in range(4):
        w, n = P.newton_solve(w); its.append(n.ksp_its)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    P.time_kernels(False)
    kt = P.kernel_times(); tm = P.timings()
    print(f"trial {trial}: {dt * 1e3:7.2f} ms per Newton iteration, its {its}, krylov {tm.krylov_ms / 4:.1f} ms; avg launch ms: "
          + ", ".join(f"{k} {v[0] / max(1, v[1]):.4f}" for k, v in kt.items() if v[1]), flush=True)
    P.close()
    if trial % 2 == 0:
        keep.append(torch.empty(int(3e8) + trial * 12345, dtype=torch.uint8, device="cuda"))     # shift the next instance's allocations
