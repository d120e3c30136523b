"""round 5: the 4-rank cavity / 3-rank duct cases of the partitioned tests through the team transport, with the library of the
tree this script is started from (PYTHONPATH decides): stokes / newton iteration counts and errors against the serial solve.
usage: python scripts/gpu_r5_teamcheck.py [KEY=VALUE ...]"""
import os, sys
root = os.environ.get("SNS_TREE", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT, _lib
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team
print("library", _lib.LIB_PATH, flush=True)
opts = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    opts[k] = float(v) if "." in v else int(v)
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for kind, nranks in (("cavity", 4), ("duct", 3), ("duct", 2)):
    if kind == "duct":
        m = M.duct_mesh((24, 6, 6), 4.0, jitter=0.1); mask, g = B.duct_bcs(m).flatten()
    else:
        m = M.cavity_mesh(12, jitter=0.1); mask, g = B.cavity_bcs(m).flatten()
    Ps = FlowProblem(m, (mask, g), reynolds=12.0)
    Us, rs = Ps.stokes_solve(); ws, ns = Ps.newton_solve(Us.clone())
    Us, ws = Us.cpu().numpy(), ws.cpu().numpy(); Ps.close()
    owner = PT.rcb_partition(m.points, nranks)
    team = Team(nranks)
    def work(rank, team):
        part = PT.build_local_part(m, mask, g, owner, rank, nranks)
        P = FlowProblem(part.mesh, (part.bc_mask, part.bc_val), reynolds=12.0, part=part, group=team, **opts)
        U, r = P.stokes_solve(); c = P.counters(); w, n = P.newton_solve(U.clone())
        out = (part, U.cpu().numpy(), r, w.cpu().numpy(), n, P.cycle(), c)
        P.close(); return out
    outs = team.run(work); team.close()
    Ug, wg = np.zeros(m.num_dofs), np.zeros(m.num_dofs)
    for part, U, r, w, n, cyc, c in outs:
        gd = (4 * part.l2g[:part.n_owned, None] + np.arange(4)[None]).ravel()
        Ug[gd], wg[gd] = U[:4 * part.n_owned], w[:4 * part.n_owned]
    o = outs[0]
    print(f"{kind} x{nranks} {opts}: serial stokes {rs.its} newton ksp {ns.ksp_its} | team stokes {o[2].its} (rnorm {o[2].rnorm:.2e}) newton ksp {o[4].ksp_its} "
          f"err_stokes {rel(Ug, Us):.2e} err_newton {rel(wg, ws):.2e} cycle {[(x['kind'], x['pre'], x['post']) for x in o[5]]} exch {o[6]['exchanges']} ar {o[6]['allreduces']}", flush=True)
