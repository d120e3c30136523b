"""Round 3 (HARNESS build): the restriction kernel that also does the next level's first sweep against the two separate launches
(SNS_NO_RESTRICT_FUSE=1): V-cycle and Krylov solution must be bitwise the same; Krylov time per iteration of both."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cases = (("duct 300x75x75", M.duct_mesh((300, 75, 75), 4.0), 200.0), ("slab share 38x75x75", M.duct_mesh((38, 75, 75), 0.5), 200.0),
         ("delaunay channel 1/28", M.delaunay_channel_mesh(28), 50.0), ("cavity 24", M.cavity_mesh(24), 100.0))
for name, m, Re in cases:
    bcs = B.cavity_bcs(m) if "cavity" in name else (B.channel_bcs(m, *B.two_stream_profiles(0.5)) if "channel" in name else B.duct_bcs(m))
    out = {}
    for rep in range(2):
        for sep in (1, 0):
            if sep: os.environ["SNS_NO_RESTRICT_FUSE"] = "1"
            else: os.environ.pop("SNS_NO_RESTRICT_FUSE", None)
            P = FlowProblem(m, bcs, reynolds=Re)
            U, r0 = P.stokes_solve()
            F = P.zeros(); P.jacobian(U, "ns", residual_out=F); P.pc_setup()
            g = torch.Generator(device="cuda").manual_seed(5)
            r = torch.randn(P.ndof, dtype=torch.float64, device="cuda", generator=g)
            z = P.pc_apply(r).cpu().numpy()
            y, k = P.krylov_solve(F); torch.cuda.synchronize()
            t0 = time.time(); y, k = P.krylov_solve(F); torch.cuda.synchronize(); dt = time.time() - t0
            out[sep] = (z, k.its, y.cpu().numpy(), dt / max(1, k.its), r0.its)
            P.close()
        print(f"{name}: per iteration separate {1e3 * out[1][3]:.4f} ms, fused {1e3 * out[0][3]:.4f} ms ({100 * (out[0][3] / out[1][3] - 1):+.1f} %); V-cycle bitwise equal "
              f"{np.array_equal(out[0][0], out[1][0])}; its {out[1][1]} / {out[0][1]}; solution equal {np.array_equal(out[0][2], out[1][2])}", flush=True)
