"""Round 3: level-2 / deep-level sweep counts on the 2.2 M-tet body-centred Delaunay channel (what the unstructured tier of
amg_nu_scale_with_size was derived from; see also gpu_r3_tierA.py, gpu_r3_hier.py)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.delaunay_channel_mesh(36, lattice="bcc")
bcs = B.channel_bcs(m, *B.two_stream_profiles(0.5))
P = FlowProblem(m, bcs, reynolds=50.0)
for kw in (dict(), dict(amg_nu_l2=8, amg_nu_deep=4), dict(amg_nu_l2=10, amg_nu_deep=8), dict(amg_nu_coarse=6, amg_nu_l2=8, amg_nu_deep=4)):
    P.set_options(amg_nu_coarse=4, amg_nu_l2=6, amg_nu_deep=2); P.set_options(**kw)
    t0 = time.time(); U, r = P.stokes_solve(); w, n = P.newton_solve(U.clone()); torch.cuda.synchronize()
    print(f"bcc 1/36 {m.num_tets} tets OPTS {kw}: stokes its {r.its} newton {n.its} its ksp/step {n.ksp_its / n.its:.1f} total {time.time() - t0:.2f}s levels {P.timings().amg_levels}", flush=True)
P.close()
