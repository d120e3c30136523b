import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells, jit = (200, 50, 50), 0.2
Re = 200.0 * cells[1] / 75.0
m = M.duct_mesh(cells, 4.0, jitter=jit)
P = FlowProblem(m, B.duct_bcs(m), reynolds=Re, amg_f32_matrix=1)
U, res = P.stokes_solve()
print("stokes", res)
F = P.zeros()
P.jacobian(U, "ns", residual_out=F)
for mr in (30, 60, 120, 200):
    P.set_options(ksp_type="fgmres", gmres_restart=mr, ksp_max_it=1000, monitor=0)
    P.pc_setup(); P.reset_timings()
    y, r = P.krylov_solve(F)
    print(f"fgmres m={mr}: its {r.its} reason {r.reason} rnorm {r.rnorm:.2e} krylov {P.timings().krylov_ms:.1f} ms", flush=True)
P.set_options(ksp_type="fgmres", gmres_restart=30, ksp_max_it=90, monitor=1)
y, r = P.krylov_solve(F)
