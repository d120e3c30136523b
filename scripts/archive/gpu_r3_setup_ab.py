"""Round 3 (HARNESS build: make -C .../csrc clean && make HARNESS=1): k_lp_copies16 with M accumulated from the registers (LDS accumulators)
against the same kernel with every row forced through the one-block-per-step loops (SNS_AP_GENERIC=1) -- the V-cycle must be bitwise
the same; setup time of both."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
for name, m, bcs in (("duct 300x75x75", None, None), ("delaunay channel 1/28", M.delaunay_channel_mesh(28), None), ("cavity 24", M.cavity_mesh(24), None)):
    if m is None:
        m = M.duct_mesh((300, 75, 75), 4.0); bcs = B.duct_bcs(m)
    elif "cavity" in name: bcs = B.cavity_bcs(m)
    else: bcs = B.channel_bcs(m, *B.two_stream_profiles(0.5))
    out = {}
    for gen in (1, 0):
        if gen: os.environ["SNS_AP_GENERIC"] = "1"
        else: os.environ.pop("SNS_AP_GENERIC", None)
        P = FlowProblem(m, bcs, reynolds=100.0)
        U, r0 = P.stokes_solve()
        F = P.zeros(); P.jacobian(U, "ns", residual_out=F)
        P.pc_setup(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.time(); P.pc_setup(); torch.cuda.synchronize(); ts.append(time.time() - t0)
        g = torch.Generator(device="cuda").manual_seed(5)
        r = torch.randn(P.ndof, dtype=torch.float64, device="cuda", generator=g)
        z = P.pc_apply(r).cpu().numpy()
        y, k = P.krylov_solve(F)
        out[gen] = (z, k.its, y.cpu().numpy(), min(ts), r0.its)
        P.close()
    print(f"{name}: setup generic {1e3 * out[1][3]:.2f} ms, registers + LDS {1e3 * out[0][3]:.2f} ms; V-cycle bitwise equal {np.array_equal(out[0][0], out[1][0])}; "
          f"its {out[1][1]} / {out[0][1]}, stokes {out[1][4]} / {out[0][4]}; solution equal {np.array_equal(out[0][2], out[1][2])}", flush=True)
