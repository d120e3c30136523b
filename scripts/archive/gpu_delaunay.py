"""Solver behaviour on a larger unstructured (Delaunay) duct mesh: iteration counts and timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Re = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
t0 = time.time(); m = M.delaunay_duct_mesh(n, 2.0, seed=3); t1 = time.time()
print(f"n={n}: {m.num_nodes} nodes {m.num_tets} tets, meshing {t1 - t0:.1f}s", flush=True)
for reorder in (False, True):
    mm = M.reorder_for_locality(m)[0] if reorder else m
    P = FlowProblem(mm, B.duct_bcs(mm), reynolds=Re)
    t0 = time.time(); U, r = P.stokes_solve(); torch.cuda.synchronize(); t1 = time.time()
    w, nres = P.newton_solve(U.clone()); torch.cuda.synchronize(); t2 = time.time()
    print(f"  morton={reorder}: stokes its {r.its} reason {r.reason} {t1 - t0:.2f}s | newton its {nres.its} reason {nres.reason} ksp its {nres.ksp_its} "
          f"fnorms {['%.1e' % f for f in nres.fnorms]} {t2 - t1:.2f}s levels {P.timings().amg_levels}", flush=True)
    P.close()
