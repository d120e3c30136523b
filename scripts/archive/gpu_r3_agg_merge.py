"""Round 3 experiment (negative): fold small aggregates into their neighbours on the config-4u Delaunay channel.  The SNS_AGG_MERGE hook
it drives is no longer in the library; profiles/r3_aggregate_merge_experiment.txt holds its source and the numbers."""
import sys, time, os
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.delaunay_channel_mesh(47, lattice="bcc"); bcs = B.channel_bcs(m, *B.two_stream_profiles(0.5))
for env in (None, "2,9", "3,10", "4,12", "5,16"):
    if env: os.environ["SNS_AGG_MERGE"] = env
    for sc in (0, 1):
        P = FlowProblem(m, bcs, reynolds=50.0, amg_nu_scale_with_size=sc)
        U, r = P.stokes_solve(); torch.cuda.synchronize()
        t0 = time.time(); w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"merge {env} scale {sc}: stokes {r.its} ksp/step {n.ksp_its / n.its:.1f} {1e3 * dt / n.its:.1f} ms/step rows {[L['rows'] for L in P.hierarchy()]}", flush=True)
        P.close()
