"""round 3: does a smaller block-Jacobi damping rescue the coarse-pillar first Jacobian (cell Reynolds number ~ 10) that round 2 documented as defeating
every Krylov method under the preconditioner?"""
import sys
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
nu = 0.001
for n in (16, 20, 24):
    m = M.reorder_for_locality(M.dfg_pillar_mesh(n))[0]
    for kw in (dict(), dict(amg_omega=0.5), dict(amg_omega=0.4), dict(amg_omega=0.3)):
        P = FlowProblem(m, B.dfg_bcs(m), reynolds=1.0 / nu, ksp_max_it=1500, **kw)
        U, r = P.stokes_solve()
        w, res = P.newton_solve(U.clone())
        print(f"pillar W/{n} ({m.num_tets} tets) OPTS {kw}: stokes its {r.its}; newton reason {res.reason} its {res.its} ksp its {res.ksp_its} retries {P.counters()['damping_retries']} "
              f"fnorms {[float(f'{x:.1e}') for x in res.fnorms]}", flush=True)
        P.close()
