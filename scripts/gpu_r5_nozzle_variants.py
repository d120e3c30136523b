"""round 5: Krylov iterations on the body-fitted nozzle channel against the shape of its cells (cross-section size, far-field plane
spacing) and the difference to the staircase channel against the resolution
usage: python scripts/gpu_r5_nozzle_variants.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stabilized_navier_stokes_flow_fenicsx_amd import inlet_image as II, nozzle_mesh as NM
from stabilized_navier_stokes_flow_fenicsx_amd.interpolate import interpolate_initial_guess
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
img = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inlet_PlusF_final.png")
def run(m, bc, **kw):
    P = FlowProblem(m, bc, reynolds=50.0, **kw)
    U, r = P.stokes_solve(); w, n = P.newton_solve(U.clone())
    W = w.cpu().numpy(); P.close()
    return W, r, n
fields = {}
for lc, cs, far in ((0.05, None, 2.0), (0.05, None, 1.0), (0.05, 0.0375, 2.0), (0.05, 0.05, 2.0), (0.05, 0.05, 1.0), (0.035, None, 2.0)):
    m, bc, data = NM.channel_from_image_bodyfitted(img, 0.5, lc, cross_size=cs, far=far)
    W, r, n = run(m, bc)
    f1, f2, fo = NM.inlet_fluxes(m, W)
    print(f"body-fitted lc {lc} cross {cs or lc / 2:g} far {far}: {m.num_tets} tets, stokes {r.its}, newton {n.its} its {n.ksp_its} ksp ({n.ksp_its / n.its:.1f}/step) reason {n.reason}, fluxes {f1:.4f} {f2:.4f} {fo:.4f}", flush=True)
    fields[(lc, cs, far)] = (m, W)
for cells in ((80, 20, 20), (160, 40, 40)):
    ms, bcs, _ = II.channel_from_image(img, 0.5, cells)
    Ws, rs, ns = run(ms, bcs)
    print(f"staircase {cells}: {ms.num_tets} tets, stokes {rs.its}, newton {ns.its} its {ns.ksp_its} ksp ({ns.ksp_its / ns.its:.1f}/step)", flush=True)
    for key in ((0.05, None, 2.0), (0.035, None, 2.0)):
        m, W = fields[key]
        Wb = interpolate_initial_guess(m, W, ms).reshape(-1, 4)
        for x0 in (0.75, 1.5):
            sel = ms.points[:, 0] > x0
            print(f"    vs body-fitted {key}: velocity difference for x > {x0}: {np.linalg.norm(Wb[sel, :3] - Ws.reshape(-1, 4)[sel, :3]) / np.linalg.norm(Ws.reshape(-1, 4)[sel, :3]):.3f}", flush=True)
# the two body-fitted resolutions against each other
m1, W1 = fields[(0.05, None, 2.0)]; m2, W2 = fields[(0.035, None, 2.0)]
Wi = interpolate_initial_guess(m2, W2, m1).reshape(-1, 4)
sel = m1.points[:, 0] > 0.75
print(f"body-fitted 0.05 vs 0.035 for x > 0.75: {np.linalg.norm(Wi[sel, :3] - W1.reshape(-1, 4)[sel, :3]) / np.linalg.norm(W1.reshape(-1, 4)[sel, :3]):.3f}")
