"""Round 3: does the first tier of the size-scaled sweeps (level 2: +2, deeper: +2) pay on 7-level hierarchies below 2.5 M rows?
Alternating A/B on the headline duct and the config-4u Delaunay channel (same box, same process)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cases = (("4u bcc 1/47", M.delaunay_channel_mesh(47, lattice="bcc"), 50.0), ("duct 300x75x75", M.duct_mesh((300, 75, 75), 4.0), 200.0))
for name, m, Re in cases:
    bcs = B.duct_bcs(m) if name.startswith("duct") else B.channel_bcs(m, *B.two_stream_profiles(0.5))
    for rep in range(2):
        for opts in ({}, {"amg_nu_l2": 8, "amg_nu_deep": 4}):
            P = FlowProblem(m, bcs, reynolds=Re, amg_nu_scale_with_size=0, **opts)
            U, r = P.stokes_solve(); torch.cuda.synchronize()
            P.set_options(snes_max_it=4) if hasattr(P, "set_options") else None
            t0 = time.time(); w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); dt = time.time() - t0
            print(f"{name:16s} {m.num_tets:9d} tets {str(opts):40s}: stokes {r.its} newton {n.its} its ksp/step {n.ksp_its / n.its:.1f} {1e3 * dt / n.its:.1f} ms/step", flush=True)
            P.close()
