"""round 2: the DFG 2D-1 benchmark (reference constants C_d = 5.57953523384, C_l = 0.010618948146,
DFG_2D_Validation.py:202-203) solved on the 3-D tet path: the 2-D triangulation extruded to a one-cell slab with u_z = 0
on both z planes, 3-D G-metric SUPG/PSPG/LSIC forms (consistent convection), 3-D traction functional per unit depth.
usage: python scripts/gpu_r2_dfg2d_on_3d_path.py [--literal] 2 4 8 [12 16]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, mesh2d as M2, functionals as Fn
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem

def solve_level(n, corrected=1):
    m3, (mask, g), h = M2.dfg2d_slab_problem(n)
    nu = 1e-3
    # residual entries scale with h^2 * thickness: the default snes_atol 1e-8 would stop after two digits
    P = FlowProblem(m3, (mask, g), reynolds=1.0 / nu, corrected_convection=corrected, snes_atol=1e-15, snes_rtol=1e-11,
                    snes_stol=1e-12, ksp_rtol=1e-10)
    t0 = time.time()
    U, rs = P.stokes_solve()
    U.view(-1, 4)[:, 3] *= nu                                          # unit-viscosity Stokes pressure -> NS scale
    w, rn = P.newton_solve(U.clone())
    dt = time.time() - t0
    F = Fn.boundary_traction_force(m3, w.cpu().numpy(), nu, m3.meta["tags"]["obstacle"])
    cd, cl = Fn.drag_lift_coefficients(F, Lc=0.1 * h)                  # per unit depth: Lc = D * thickness
    P.close()
    return m3.num_tets, h, rs, rn, cd, cl, dt

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    corrected = 0 if "--literal" in sys.argv else 1         # --literal: the reference's 3-D form as written (dot(u, grad(.)))
    levels = [float(a) for a in args] or [2, 4, 8]
    print("3-D form:", "consistent convection in the stabilisation (corrected_convection=1)" if corrected else "literal (as NavierStokesChannelFlow.py:220-251)")
    for n in levels:
        nt, h, rs, rn, cd, cl, dt = solve_level(n, corrected)
        print(f"level {n:g}: {nt} tets, h {h:.4f}, stokes its {rs.its}, newton its {rn.its} reason {rn.reason} ksp {rn.ksp_its}, "
              f"|F| {rn.fnorms[0]:.2e} -> {rn.fnorms[-1]:.2e}, C_d {cd:.6f} ({(cd / M2.DFG2D_CD_REF - 1) * 100:+.3f} %), C_l {cl:.6f} ({(cl / M2.DFG2D_CL_REF - 1) * 100:+.2f} %), {dt:.1f} s", flush=True)
