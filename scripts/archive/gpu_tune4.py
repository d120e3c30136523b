import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = (300, 75, 75)
m = M.duct_mesh(cells, 4.0)
bcs = B.duct_bcs(m)
for agg in (8,):
    P = FlowProblem(m, bcs, reynolds=200.0, amg_agg_size=agg)
    U, res = P.stokes_solve()
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    for nu, nuc, om in [(1, 4, 0.8), (1, 4, 0.9), (1, 4, 1.0), (1, 4, 0.7)]:
        P.set_options(amg_nu=nu, amg_nu_coarse=nuc, amg_omega=om, monitor=1)
        P.pc_setups_reset = None
        P.pc_setup(); P.set_options(monitor=0); P.reset_timings()
        y, r = P.krylov_solve(F)
        t = P.timings()
        print(f"agg {agg:2d} nu {nu} nuc {nuc} omega_cap {om}: its {r.its} reason {r.reason} krylov {t.krylov_ms:.1f} ms levels {t.amg_levels}", flush=True)
    P.close()
