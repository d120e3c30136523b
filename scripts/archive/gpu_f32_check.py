import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
for cells, jit in [((300, 75, 75), 0.0), ((200, 50, 50), 0.2)]:
    Re = 200.0 * cells[1] / 75.0
    m = M.duct_mesh(cells, 4.0, jitter=jit)
    P = FlowProblem(m, B.duct_bcs(m), reynolds=Re)
    U, res = P.stokes_solve()
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    sols = {}
    for f32 in (0, 1):
        for ksp in ("bicgstab", "fgmres", "tfqmr"):
            P.set_options(ksp_type=ksp, amg_f32_matrix=f32)
            P.pc_setup(); P.reset_timings(); P.time_kernels(True)
            y, r = P.krylov_solve(F)
            t = P.timings(); kt = P.kernel_times()
            sols[(f32, ksp)] = y
            print(f"{cells} jit {jit} f32 {f32} {ksp:8s}: its {r.its:4d} reason {r.reason} rnorm {r.rnorm:.2e} krylov {t.krylov_ms:7.1f} ms  jacobi avg {kt['jacobi'][0]/max(1,kt['jacobi'][1]):.4f} ms", flush=True)
    ref = sols[(0, "bicgstab")]
    for k, v in sols.items():
        print("   ", k, "rel diff vs f64 bicgstab %.2e" % float((v - ref).norm() / ref.norm()))
    P.close()
