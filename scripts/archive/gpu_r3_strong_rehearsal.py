"""round 3: bench.py's STRONG layout (the one 300x75x75 mesh in N x-slabs) as N threads on ONE GPU over the team transport, with the
production two-stream halo choreography (SNS_TEAM_OVERLAP=1): iteration counts and per-solve collective counters of the partitioned
solver after the round-3 changes (fused post-sweep: one level-1 exchange instead of the fine-level halo of the corrected iterate).
Timings are not meaningful (the ranks share one GPU)."""
import sys, os, time
os.environ["SNS_TEAM_OVERLAP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team
cells = (300, 75, 75) if len(sys.argv) < 3 else tuple(int(c) for c in sys.argv[2].split(","))
for N in [int(a) for a in sys.argv[1].split(",")]:
    if N == 1:
        m = M.duct_mesh(cells, 4.0)
        P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, snes_max_it=1)
        U, r = P.stokes_solve(); w, n1 = P.newton_solve(U.clone()); w, n2 = P.newton_solve(w)
        print(f"N=1: stokes its {r.its} newton ksp its {n1.ksp_its},{n2.ksp_its} levels {P.timings().amg_levels}", flush=True)
        P.close(); del P, m
        continue
    team = Team(N)

    def work(rank, team):
        part = PT.duct_slab_part(cells, 4.0, rank, N)
        P = FlowProblem.from_part(part, group=team, reynolds=200.0, snes_max_it=1)
        U, r = P.stokes_solve()
        w, n1 = P.newton_solve(U.clone())
        c = P.counters()
        w, n2 = P.newton_solve(w)
        out = (part.n_owned, r.its, n1.ksp_its, n2.ksp_its, n2.fnorms[-1], P.timings().amg_levels, c["allreduces"], c["exchanges"])
        P.close()
        return out

    t0 = time.time()
    outs = team.run(work)
    team.close()
    o = outs[0]
    print(f"N={N}: stokes its {o[1]} newton ksp its {o[2]},{o[3]} |F| {o[4]:.2e} levels {o[5]}; first Newton solve: {o[6]} all-reduces, "
          f"{o[7]} halo exchanges = {o[6] / max(1, o[2]):.1f} / {o[7] / max(1, o[2]):.1f} per iteration  (wall {time.time() - t0:.0f}s)", flush=True)
