"""round 4: bench.py's STRONG layout (the one 300x75x75 mesh in N x-slabs) as N threads on ONE GPU over the team transport, with the
production two-stream halo choreography (SNS_TEAM_OVERLAP=1): iteration counts, the cycle as run (smoother kinds / sweeps) and the
per-solve collective counters of the partitioned solver with the round-4 hierarchy (aggregate-block smoothing on the latency-bound
levels, dense coarsest level in the replicated tail).  Timings are not meaningful (the ranks share one GPU).
usage: python scripts/gpu_r4_strong_rehearsal.py 1,2,4,8 [cells] [KEY=VALUE ...]"""
import sys, os, time
os.environ["SNS_TEAM_OVERLAP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team
args = sys.argv[1:]
opts = {}
for a in [a for a in args if "=" in a]:
    k, v = a.split("=")
    opts[k] = float(v) if "." in v else int(v)
args = [a for a in args if "=" not in a]
cells = (300, 75, 75) if len(args) < 2 else tuple(int(c) for c in args[1].split(","))
print("options", opts, flush=True)
for N in [int(a) for a in args[0].split(",")]:
    if N == 1:
        m = M.duct_mesh(cells, 4.0)
        P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, snes_max_it=1, **opts)
        U, r = P.stokes_solve(); w, n1 = P.newton_solve(U.clone()); w, n2 = P.newton_solve(w)
        print(f"N=1: stokes its {r.its} newton ksp its {n1.ksp_its},{n2.ksp_its} rows {[h['rows'] for h in P.hierarchy()]} "
              f"cycle {[(c['kind'], c['pre'], c['post']) for c in P.cycle()]}", flush=True)
        P.close(); del P, m
        continue
    team = Team(N)

    def work(rank, team):
        part = PT.duct_slab_part(cells, 4.0, rank, N)
        P = FlowProblem.from_part(part, group=team, reynolds=200.0, snes_max_it=1, **opts)
        U, r = P.stokes_solve()
        w, n1 = P.newton_solve(U.clone())
        c = P.counters()
        w, n2 = P.newton_solve(w)
        out = (part.n_owned, r.its, n1.ksp_its, n2.ksp_its, n2.fnorms[-1], [h["rows"] for h in P.hierarchy()],
               c["allreduces"], c["exchanges"], [(x["kind"], x["pre"], x["post"]) for x in P.cycle()])
        P.close()
        return out

    t0 = time.time()
    outs = team.run(work)
    team.close()
    o = outs[0]
    print(f"N={N}: stokes its {o[1]} newton ksp its {o[2]},{o[3]} |F| {o[4]:.2e} rows(rank 0) {o[5]} cycle {o[8]}; first Newton solve: "
          f"{o[6]} all-reduces, {o[7]} halo exchanges = {o[6] / max(1, o[2]):.1f} / {o[7] / max(1, o[2]):.1f} per iteration  "
          f"(wall {time.time() - t0:.0f}s)", flush=True)
