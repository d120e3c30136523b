"""round 3: the FIRST Newton step (Jacobian at the Stokes solution) on coarse ducts at Re 200 (cell Reynolds number 8-16): damping per level,
iterations of the first attempt / retry, and the effect of a hand-set damping"""
import sys
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
for cells in [tuple(int(c) for c in a.split(",")) for a in (sys.argv[1:] or ["160,40,40"])]:
    m = M.duct_mesh(cells, 4.0)
    for kw in (dict(), dict(amg_omega=0.6), dict(amg_omega=0.5), dict(amg_retry_damping=0, ksp_max_it=800)):
        P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, **kw)
        U, r = P.stokes_solve()
        F = P.zeros()
        P.jacobian(U, "ns", residual_out=F)
        P.set_options(monitor=1)
        P.pc_setup()
        P.set_options(monitor=0)
        y, k = P.krylov_solve(F)
        c = P.counters()
        print(f"CELLS {cells} OPTS {kw}: stokes its {r.its}; first Jacobian: its {k.its} reason {k.reason} retries {c['damping_retries']} first-attempt reason {c['first_attempt_reason']}", flush=True)
        P.close()
