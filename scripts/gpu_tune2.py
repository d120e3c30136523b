import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
for cells, jit in [((300, 75, 75), 0.0), ((200, 50, 50), 0.2)]:
    Re = 200.0 * cells[1] / 75.0
    m = M.duct_mesh(cells, 4.0, jitter=jit)
    P = FlowProblem(m, B.duct_bcs(m), reynolds=Re, monitor=0)
    U, res = P.stokes_solve()
    print(cells, "jitter", jit, "stokes", res, flush=True)
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    for ksp, om in [("bicgstab", 0.8), ("fgmres", 0.8), ("bicgstab", 1.0), ("bicgstab", 0.7)]:
        P.set_options(ksp_type=ksp, amg_omega=om, monitor=1 if (ksp == "bicgstab" and om == 1.0) else 0, ksp_max_it=300)
        P.pc_setup()
        P.set_options(monitor=0)
        P.reset_timings()
        y, r = P.krylov_solve(F)
        t = P.timings()
        print(f"   {ksp:8s} omega_cap {om}: its {r.its:4d} reason {r.reason} krylov {t.krylov_ms:8.1f} ms pc_setup {t.pc_setup_ms:.1f}", flush=True)
    P.close()
