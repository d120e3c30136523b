import sys, os, time
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = tuple(int(c) for c in sys.argv[1].split(","))
part = PT.duct_slab_part(cells, 4.0, 0, 1)
P = FlowProblem(part.mesh, (part.bc_mask, part.bc_val), reynolds=200.0, snes_max_it=1)
for kw in (dict(), dict(amg_nu_l2=8, amg_nu_deep=6), dict(amg_nu_l2=10, amg_nu_deep=10), dict(amg_nu_coarse=5, amg_nu_l2=8, amg_nu_deep=6)):
    P.set_options(amg_nu_coarse=4, amg_nu_l2=6, amg_nu_deep=2)
    P.set_options(**kw)
    t0 = time.time(); U, r = P.stokes_solve(); torch.cuda.synchronize(); t1 = time.time()
    w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); t2 = time.time()
    w, n2 = P.newton_solve(w); torch.cuda.synchronize(); t3 = time.time()
    print(f"{cells} OPTS {kw}: stokes its {r.its} ({t1-t0:.1f}s) newton ksp its {n.ksp_its},{n2.ksp_its} {t2-t1:.2f}s,{t3-t2:.2f}s", flush=True)
P.close()
