"""DFG pillar case (h = W/32) under several solver option sets: iteration counts on an unstructured mesh."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
n = int(os.environ.get("DFG_N", "32"))
m = M.reorder_for_locality(M.dfg_pillar_mesh(n))[0]
bc = B.dfg_bcs(m)
for spec in sys.argv[1:] or [""]:
    kw = {}
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("="); kw[k] = float(v) if "." in v else (v if not v.lstrip("-").isdigit() else int(v))
    P = FlowProblem(m, bc, reynolds=1000.0, ksp_max_it=3000, **kw)
    t0 = time.time(); U, r = P.stokes_solve(); w, res = P.newton_solve(U.clone()); torch.cuda.synchronize(); t1 = time.time()
    print(f"{spec or 'default':40s} stokes {r.its:4d} newton its {res.its} reason {res.reason} ksp {res.ksp_its:5d} levels {P.timings().amg_levels} {t1 - t0:.2f}s", flush=True)
    P.close()
