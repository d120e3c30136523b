import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.duct_mesh((300, 75, 75), 4.0)
P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0)
U, res = P.stokes_solve()
F = P.zeros()
P.jacobian(U, "ns", residual_out=F)
for rep in range(2):
    P.pc_setup(); P.reset_timings()
    y, r = P.krylov_solve(F)
    print(f"SNS_NU_DEEP={os.environ.get('SNS_NU_DEEP')}: its {r.its} reason {r.reason} krylov {P.timings().krylov_ms:.1f} ms", flush=True)
