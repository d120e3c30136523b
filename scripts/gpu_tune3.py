import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = (300, 75, 75)
m = M.duct_mesh(cells, 4.0)
P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0)
U, res = P.stokes_solve()
F = P.zeros()
P.jacobian(U, "ns", residual_out=F)
for nu, nuc, agg in [(1, 3, 8), (1, 4, 8), (1, 5, 8), (1, 6, 8), (1, 8, 8), (1, 12, 8)]:
    P.set_options(amg_nu=nu, amg_nu_coarse=nuc)
    P.pc_setup(); P.reset_timings()
    y, r = P.krylov_solve(F)
    print(f"nu {nu} nu_coarse {nuc}: its {r.its} reason {r.reason} krylov {P.timings().krylov_ms:.1f} ms", flush=True)
