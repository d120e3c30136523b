"""N-rank solver logic on ONE GPU through the in-process team transport."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B, partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team

cells = eval(sys.argv[1]) if len(sys.argv) > 1 else (32, 8, 8)
OPTS = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("SNS_TEAM_OPTS", "").split(",") if kv)}
Re = 200.0 * cells[1] / 75.0
m = M.duct_mesh(cells, 4.0)
mask, g = B.duct_bcs(m).flatten()
Ps = FlowProblem(m, (mask, g), reynolds=Re, **OPTS)
Us, rs = Ps.stokes_solve()
ws, ns = Ps.newton_solve(Us.clone())
print("serial: stokes its", rs.its, "newton", ns.its, ns.reason, "ksp", ns.ksp_its, flush=True)
for nr in (2, 4, 8):
    owner = PT.rcb_partition(m.points, nr)
    team = Team(nr)
    def work(rank, team):
        part = PT.build_local_part(m, mask, g, owner, rank, nr)
        P = FlowProblem(part.mesh, (part.bc_mask, part.bc_val), reynolds=Re, part=part, group=team, **OPTS)
        U, r = P.stokes_solve()
        w, n = P.newton_solve(U.clone())
        res = (part, U.cpu().numpy(), r, w.cpu().numpy(), n, P.timings().amg_levels)
        P.close()
        return res
    t0 = time.time()
    outs = team.run(work)
    Ug = np.zeros(m.num_dofs); wg = np.zeros(m.num_dofs)
    for part, U, r, w, n, lev in outs:
        gd = (4 * part.l2g[:part.n_owned, None] + np.arange(4)[None]).ravel()
        Ug[gd] = U[:4 * part.n_owned]; wg[gd] = w[:4 * part.n_owned]
    r0, n0 = outs[0][2], outs[0][4]
    print(f"ranks {nr}: stokes its {r0.its} reason {r0.reason} err {np.linalg.norm(Ug-Us.cpu().numpy())/np.linalg.norm(Us.cpu().numpy()):.2e};"
          f" newton {n0.its} {n0.reason} ksp {n0.ksp_its} err {np.linalg.norm(wg-ws.cpu().numpy())/np.linalg.norm(ws.cpu().numpy()):.2e} levels {outs[0][5]} ({time.time()-t0:.1f}s)", flush=True)
    team.close()
