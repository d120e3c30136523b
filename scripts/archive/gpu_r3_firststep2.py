import sys
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.duct_mesh((160, 40, 40), 4.0)
for kw in (dict(), dict(amg_omega=0.5), dict(amg_omega=0.55), dict(amg_retry_stall_its=60)):
    P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, **kw)
    U, r = P.stokes_solve()
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    P.set_options(monitor=1)
    y, k = P.krylov_solve(F)
    P.set_options(monitor=0)
    print(f"OPTS {kw}: its {k.its} reason {k.reason} retries {P.counters()['damping_retries']}", flush=True)
    P.close()
