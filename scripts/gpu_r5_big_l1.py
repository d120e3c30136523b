"""round 5: the level-1 schedule (and the level-2 / deep ones) on LARGE single-GPU meshes -- iterations and time per Newton
iteration of the 81 M-tet duct (VERDICT r4 item 7's stretch: <= 50 iterations) against amg_bnu_l1 / amg_bnu_l2 / amg_bnu_deep
usage: python scripts/gpu_r5_big_l1.py 600,150,150"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = tuple(int(c) for c in sys.argv[1].split(","))
part = PT.duct_slab_part(cells, 4.0, 0, 1)
P = FlowProblem(part.mesh, (part.bc_mask, part.bc_val), reynolds=200.0, snes_max_it=1)
base = dict(amg_bnu_l1=3, amg_bnu_l2=4, amg_bnu_deep=2, amg_nu_l1_pre=0, amg_nu_l1_post=0)
for kw in (dict(), dict(amg_bnu_l1=4), dict(amg_bnu_l1=5), dict(amg_nu_l1_pre=2, amg_nu_l1_post=4), dict(amg_bnu_l1=4, amg_bnu_l2=5),
           dict(amg_bnu_l1=4, amg_bnu_l2=5, amg_bnu_deep=3)):
    P.set_options(**base)
    P.set_options(**kw)
    t0 = time.time(); U, r = P.stokes_solve(); torch.cuda.synchronize(); t1 = time.time()
    w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); t2 = time.time()
    w, n2 = P.newton_solve(w); torch.cuda.synchronize(); t3 = time.time()
    w, n3 = P.newton_solve(w); torch.cuda.synchronize(); t4 = time.time()
    cyc = [(c["kind"], c["pre"], c["post"]) for c in P.cycle()]
    print(f"{cells} {kw}: stokes its {r.its} ({t1-t0:.2f}s) newton ksp its {n.ksp_its},{n2.ksp_its},{n3.ksp_its} {t2-t1:.3f}s,{t3-t2:.3f}s,{t4-t3:.3f}s cycle {cyc}", flush=True)
P.close()
