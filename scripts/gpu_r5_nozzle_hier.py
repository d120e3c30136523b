"""round 5: where do the iterations of the body-fitted nozzle channel come from?  The hierarchy (rows per level = nodes per aggregate)
and the Krylov iterations per Newton step against the spacing of the node planes: the reference's graded size field (default),
uniform planes at the cross-section's size (isotropic prisms), uniform planes at twice that.
usage: python scripts/gpu_r5_nozzle_hier.py [lc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stabilized_navier_stokes_flow_fenicsx_amd import inlet_image as II, nozzle_mesh as NM
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
img = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inlet_PlusF_final.png")
lc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
opts = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    opts[k] = float(v) if "." in v else int(v)
def run(name, m, bc):
    P = FlowProblem(m, bc, reynolds=50.0, **opts)
    U, r = P.stokes_solve(); w, n = P.newton_solve(U.clone())
    rows = [h["rows"] for h in P.hierarchy()]
    cyc = [(c["kind"], c["pre"], c["post"]) for c in P.cycle()]
    om = [round(h.get("omega", 0.0), 3) for h in P.hierarchy()]
    P.close()
    print(f"{name}: {m.num_tets} tets, {m.num_nodes} nodes, stokes {r.its}, newton {n.its} its {n.ksp_its} ksp ({n.ksp_its / n.its:.1f}/step) reason {n.reason}\n"
          f"    rows {rows}  ratios {[round(rows[i] / rows[i + 1], 2) for i in range(len(rows) - 1)]}\n    cycle {cyc} omega {om}", flush=True)
orig = NM.size_along_x
m, bc, _ = NM.channel_from_image_bodyfitted(img, 0.5, lc)
run("graded planes (default)", m, bc)
for f, name in ((0.75, "uniform planes at the cross-section size"), (1.5, "uniform planes at 2 x the cross-section size")):
    NM.size_along_x = lambda x, lc_, x_extrude=0.5, growth=0.35, far=2.0, f=f: np.full_like(np.asarray(x, dtype=np.float64), f * lc_)
    m, bc, _ = NM.channel_from_image_bodyfitted(img, 0.5, lc)
    run(name, m, bc)
NM.size_along_x = orig
for xe in (0.25, 0.1):                                   # a shorter nozzle (the reference's x_extrude is 0.5, image2gmsh3D.py:193)
    m, bc, _ = NM.channel_from_image_bodyfitted(img, 0.5, lc, x_extrude=xe)
    run(f"graded planes, nozzle length {xe}", m, bc)
ms, bcs, _ = II.channel_from_image(img, 0.5, (80, 20, 20))
run("staircase 80 x 20 x 20", ms, bcs)
