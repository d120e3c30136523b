"""Reynolds-number continuation on a coarse DFG pillar mesh where the direct Newton solve at Re = 1000 fails."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B, functionals as Fn
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m = M.reorder_for_locality(M.dfg_pillar_mesh(n))[0]
P = FlowProblem(m, B.dfg_bcs(m), reynolds=1000.0, ksp_max_it=3000)
w, r = P.stokes_solve()
for Re in [float(a) for a in sys.argv[2:]] or [250.0, 500.0, 1000.0]:
    P.set_options(reynolds=Re)
    w2, res = P.newton_solve(w.clone())
    print(f"Re {Re:g}: newton its {res.its} reason {res.reason} ksp {res.ksp_its} |F| {res.fnorms[-1] if res.fnorms else float('nan'):.2e}", flush=True)
    if res.reason > 0:
        w = w2
f = Fn.boundary_traction_force(m, w.cpu().numpy(), 1.0 / Re, m.meta["tags"]["obstacle"])
print("C_d, C_l at the last converged Re:", Fn.drag_lift_coefficients(f))
