"""round 5: what do the reference-held constants C_d / C_l (DFG_2D_Validation.py:202-203) catch?  DFG 2D-1 on the
one-cell slab of tets through the 3-D kernels (tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path) with the form as
written, the consistent convection, and five perturbations of the stabilisation (sns_set_form_variant): C_I 36 -> 4, tau without
the 36 nu^2 G:G term, LSIC off, PSPG sign flipped, 1-point quadrature.  Per variant: C_d, C_l and whether C_d leaves the
[consistent, literal] bracket of the level.
usage: python scripts/gpu_r5_pin_variants.py [levels, default 4,8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stabilized_navier_stokes_flow_fenicsx_amd import functionals as Fn, mesh2d as M2
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
NU = 1e-3
levels = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,8").split(",")]
VARIANTS = [("literal (reference as written)", 0, {}), ("consistent convection", 1, {}),
            ("C_I 36 -> 4", 0, dict(c_inverse=4.0)), ("tau without 36 nu^2 G:G", 0, dict(c_inverse=0.0)),
            ("LSIC off", 0, dict(lsic_scale=0.0)), ("PSPG sign flipped", 0, dict(pspg_sign=-1.0)),
            ("1-point quadrature", 0, dict(one_point_quadrature=True)),
            ("C_I 36 -> 144", 0, dict(c_inverse=144.0)), ("LSIC x 4", 0, dict(lsic_scale=4.0))]
for n in levels:
    m3, (mask, g), thick = M2.dfg2d_slab_problem(n)
    res = {}
    for name, corrected, kw in VARIANTS:
        P = FlowProblem(m3, (mask, g), reynolds=1.0 / NU, corrected_convection=corrected, snes_atol=1e-15, snes_rtol=1e-11,
                        snes_stol=1e-12, ksp_rtol=1e-10, snes_max_it=40)
        if kw:
            P.set_form_variant(**kw)
        try:
            U, rs = P.stokes_solve()
            U.view(-1, 4)[:, 3] *= NU
            w, rn = P.newton_solve(U.clone())
            W = w.cpu().numpy()
            F = Fn.boundary_traction_force(m3, W, NU, m3.meta["tags"]["obstacle"])
            cd, cl = Fn.drag_lift_coefficients(F, Lc=0.1 * thick)
            res[name] = (cd, cl, rn.reason, rn.its, rn.ksp_its)
        except Exception as e:      # noqa: BLE001
            res[name] = (float("nan"), float("nan"), -99, 0, 0)
            print("   ", name, "failed:", e)
        P.close()
    lo, hi = res["consistent convection"][0], res["literal (reference as written)"][0]
    print(f"level {n}: {m3.num_tets} tets; bracket [consistent, literal] = [{lo:.6f}, {hi:.6f}] ({100 * (lo / M2.DFG2D_CD_REF - 1):+.3f} % .. "
          f"{100 * (hi / M2.DFG2D_CD_REF - 1):+.3f} % of C_d_ref {M2.DFG2D_CD_REF})", flush=True)
    for name, corrected, kw in VARIANTS:
        cd, cl, reason, its, kits = res[name]
        out = "inside " if lo <= cd <= hi else "OUTSIDE"
        print(f"   {name:34s} C_d {cd:10.6f} ({100 * (cd / M2.DFG2D_CD_REF - 1):+8.3f} %) {out} bracket   C_l {cl:10.6f} ({100 * (cl / M2.DFG2D_CL_REF - 1):+8.2f} %)   "
              f"SNES reason {reason}, {its} its, {kits} ksp its", flush=True)
