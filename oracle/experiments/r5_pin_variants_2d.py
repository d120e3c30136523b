"""round 5, 2-D half: what do the reference-held constants C_d / C_l (DFG_2D_Validation.py:202-203) catch on the
form they were written for (the 2-D UGN form, DFG_2D_Validation.py:141-163)?  The oracle's literal restatement (oracle/forms2d.py,
sparse-LU Newton) with perturbations of the stabilisation: LSIC off / x 4, PSPG sign flipped, tau_SUNG3 = h^2/(4 nu) -> h^2/(nu) and
h^2/(16 nu), 1-point quadrature.  CPU only (the HIP path equals this oracle to 1e-12, tests/test_gpu_2d.py).
usage: python oracle/experiments/r5_pin_variants_2d.py [levels, default 2,4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import forms2d as F2
from stabilized_navier_stokes_flow_fenicsx_amd import mesh2d as M2
NU = 1e-3
levels = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,4").split(",")]
VARIANTS = [("reference form", {}), ("LSIC off", dict(lsic=0.0)), ("LSIC x 4", dict(lsic=4.0)), ("PSPG sign flipped", dict(pspg=-1.0)),
            ("tau_SUNG3 h^2/(4 nu) -> h^2/nu", dict(sung3=1.0)), ("tau_SUNG3 h^2/(4 nu) -> h^2/(16 nu)", dict(sung3=16.0)),
            ("1-point quadrature", dict(one_point=True))]
for n in levels:
    m = M2.dfg_2d_mesh(n)
    mask, g = M2.dfg2d_bcs(m).flatten()
    print(f"level {n}: {m.num_cells} triangles", flush=True)
    for name, kw in VARIANTS:
        F2.VARIANT.update(dict(lsic=1.0, pspg=1.0, sung3=4.0, one_point=False))
        F2.VARIANT.update(kw)
        t0 = time.time()
        try:
            U = F2.solve_stokes2d(m.points, m.tris, mask, g, 1.0, 0.2)
            U.reshape(-1, 4)[:, 3] *= NU
            w, info = F2.newton2d(m.points, m.tris, U, NU, mask, g, max_it=25)
            cd, cl = M2.drag_lift_2d(m, w, NU)
            print(f"   {name:38s} C_d {cd:10.6f} ({100 * (cd / M2.DFG2D_CD_REF - 1):+8.3f} %)   C_l {cl:10.6f} ({100 * (cl / M2.DFG2D_CL_REF - 1):+8.2f} %)   "
                  f"Newton {'converged' if info['converged'] else 'NOT converged'} in {info['its']} its   ({time.time() - t0:.0f} s)", flush=True)
        except Exception as e:       # noqa: BLE001
            print(f"   {name:38s} failed: {type(e).__name__}: {e}", flush=True)
