"""DFG pillar case: the reference's convective term as written, dot(u, grad(u)) = (grad u)^T u, against the corrected
(u . grad) u -- same mesh, Reynolds continuation for both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B, functionals as Fn
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, newton_with_reynolds_continuation
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
m = M.reorder_for_locality(M.dfg_pillar_mesh(n))[0]
for corrected in (0, 1):
    P = FlowProblem(m, B.dfg_bcs(m), reynolds=1000.0, ksp_max_it=3000, corrected_convection=corrected)
    U, r = P.stokes_solve()
    w, res = newton_with_reynolds_continuation(P, U.clone(), verbose=True)
    wh = w.cpu().numpy()
    cd, cl = Fn.drag_lift_coefficients(Fn.boundary_traction_force(m, wh, 0.001, m.meta["tags"]["obstacle"]))
    W4 = wh.reshape(-1, 4)
    near = lambda x, y, z: W4[np.argmin(((m.points - np.array([x, y, z])) ** 2).sum(axis=1)), 3]
    print(f"corrected_convection={corrected}: reason {res.reason} C_d {cd:.4f} C_l {cl:.5f} dp {near(0.45, 0.2, 0.205) - near(0.55, 0.2, 0.205):.4f}", flush=True)
    P.close()
