import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
for cells, L in [((128, 16, 16), 4.0), ((64, 32, 32), 4.0), ((64, 16, 16), 4.0), ((256, 16, 16), 4.0)]:
    h = (L / cells[0], 1.0 / cells[1])
    Re = 2.67 / max(h)
    m = M.duct_mesh(cells, L)
    P = FlowProblem(m, B.duct_bcs(m), reynolds=Re, ksp_max_it=400)
    U, res = P.stokes_solve()
    w, r = P.newton_solve(U.clone())
    print(f"cells {cells} aspect {h[0]/h[1]:.2f} Re {Re:.0f}: stokes its {res.its} reason {res.reason}; newton its {r.its} reason {r.reason} ksp {r.ksp_its}", flush=True)
    P.close()
