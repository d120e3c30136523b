"""Node ordering A/B: lexicographic (x slowest) vs Morton order of the same 10.1 M-tet duct."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
from stabilized_navier_stokes_flow_fenicsx_amd._lib import check
m0 = M.duct_mesh((300, 75, 75), 4.0)
t0 = time.time(); m1, perm = M.reorder_for_locality(m0); print("reorder %.1fs" % (time.time() - t0), flush=True)
for name, m in (("lexicographic", m0), ("morton", m1)):
    P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, snes_max_it=1)
    U, r = P.stokes_solve()
    w, n = P.newton_solve(U.clone())
    P.reset_timings()
    torch.cuda.synchronize(); t0 = time.time()
    w2, n2 = P.newton_solve(w)
    torch.cuda.synchronize(); dt = time.time() - t0
    P.jacobian(w, "ns"); P.pc_setup()
    out = []
    for which in (3, 1):
        ms = (C.c_double * 2)()
        check(P.lib.sns_bench_variants(P.h, which, 4, 10, ms))
        out.append(ms[0])
    asm = P.bench_assemble(w, "ns", 5)
    print(f"{name}: stokes its {r.its} newton ksp its {n.ksp_its},{n2.ksp_its} step {dt*1e3:.1f} ms  fp64 ax {out[0]:.4f} ms  f32 jacobi {out[1]:.4f} ms  assemble {asm:.3f} ms", flush=True)
    P.close()
