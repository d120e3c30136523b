import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = (300, 75, 75)
m = M.duct_mesh(cells, 4.0)
bcs = B.duct_bcs(m)
for cs in (32, 100, 160):
    P = FlowProblem(m, bcs, reynolds=200.0, amg_coarse_size=cs)
    U, res = P.stokes_solve()
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    for nuc in (4, 3):
        P.set_options(amg_nu_coarse=nuc)
        P.pc_setup(); P.reset_timings()
        y, r = P.krylov_solve(F)
        t = P.timings()
        P.reset_timings(); P.pc_setup(); ts = P.timings().pc_setup_ms
        print(f"coarse_size {cs} nuc {nuc}: its {r.its} reason {r.reason} krylov {t.krylov_ms:.1f} ms levels {t.amg_levels} pc_setup {ts:.1f} ms", flush=True)
    P.close()
