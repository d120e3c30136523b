"""round 2: one more level of the DFG-2D series (level 32, ~5.4 M triangles) for the table in DESIGN.md section 5."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import mesh2d as M2
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
NU = 1e-3
for n in (24, 32):
    t = time.time(); m = M2.dfg_2d_mesh(n, smooth=1); tm = time.time() - t
    mask, g = M2.dfg2d_bcs(m).flatten()
    P = FlowProblem(m, (mask, g), reynolds=1.0 / NU)
    U, res = P.stokes_solve()
    U.view(-1, 4)[:, 3] *= NU
    w, nres = P.newton_solve(U)
    cd, cl = M2.drag_lift_2d(m, w.cpu().numpy(), NU)
    print(f"level {n}: {m.num_cells} triangles (meshed in {tm:.0f} s), C_d {cd:.6f} ({100 * (cd / M2.DFG2D_CD_REF - 1):+.4f} %), "
          f"C_l {cl:.6f} ({100 * (cl / M2.DFG2D_CL_REF - 1):+.3f} %), stokes its {res.its}, Newton {nres.its} its reason {nres.reason}, "
          f"{nres.ksp_its} ksp its, {nres.seconds:.2f} s", flush=True)
    P.close()
