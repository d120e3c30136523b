"""Rehearse bench.py's weak-scaling layout on ONE GPU: N ranks = N threads over the in-process team transport,
each meshing its own ~10 M-tet x-slab of the N^(1/3)-refined duct.  Reports iteration counts (timings are not
meaningful: the ranks share one GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team
base = (300, 75, 75) if len(sys.argv) < 3 else tuple(int(c) for c in sys.argv[2].split(","))
extra = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else [])}
for N in [int(a) for a in sys.argv[1].split(",")]:
    sc = float(N) ** (1.0 / 3.0)
    cells = tuple(int(round(c * sc)) for c in base)
    team = Team(N)

    def work(rank, team):
        t0 = time.time()
        part = PT.duct_slab_part(cells, 4.0, rank, N)
        t1 = time.time()
        P = FlowProblem.from_part(part, group=team, reynolds=200.0, snes_max_it=1, **extra)
        U, r = P.stokes_solve()
        w, n1 = P.newton_solve(U.clone())
        w, n2 = P.newton_solve(w)
        out = (part.n_owned, part.mesh.num_tets, t1 - t0, r.its, n1.ksp_its, n2.ksp_its, n1.fnorms[-1], n2.fnorms[-1], P.timings().amg_levels)
        P.close()
        return out

    t0 = time.time()
    outs = team.run(work)
    team.close()
    print(f"N={N} cells={cells} wall {time.time() - t0:.1f}s", flush=True)
    for o in outs:
        print("   owned nodes %d local tets %d part-build %.1fs stokes its %d newton ksp its %d,%d fnorm %.3e,%.3e levels %d" % o, flush=True)
