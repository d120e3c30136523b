"""round 2: FGMRES under the partitioned AMG hierarchy (team transport, 1/2/4 ranks) on a 110 k-tet duct."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team
m = M.duct_mesh((72, 16, 16), 4.0, jitter=0.1)
mask, g = B.duct_bcs(m).flatten()
print("tets", m.num_tets)
for ksp in ("bicgstab", "fgmres"):
    P = FlowProblem(m, (mask, g), reynolds=50.0, ksp_type=ksp)
    U, r = P.stokes_solve(); w, n = P.newton_solve(U.clone())
    print(ksp, "serial: stokes", r.its, r.reason, "newton", n.its, n.reason, n.ksp_its, flush=True)
    P.close()
    for nr in (2, 4):
        owner = PT.rcb_partition(m.points, nr)
        team = Team(nr)
        def work(rank, team):
            part = PT.build_local_part(m, mask, g, owner, rank, nr)
            P = FlowProblem(part.mesh, (part.bc_mask, part.bc_val), reynolds=50.0, ksp_type=ksp, part=part, group=team, ksp_max_it=400)
            U, r = P.stokes_solve(); w, n = P.newton_solve(U.clone())
            P.close()
            return (r.its, r.reason, n.its, n.reason, n.ksp_its)
        out = team.run(work); team.close()
        print(ksp, nr, "ranks:", out[0], flush=True)
