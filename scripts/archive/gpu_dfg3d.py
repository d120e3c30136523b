"""DFG 3D-1Z style benchmark of Validation_Flow/DFG_3D_Validation.py on Delaunay meshes of the pillar channel:
drag / lift coefficients vs the mesh size (literature for the 3D-1Z cylinder case: C_d 6.05-6.25, C_l 0.008-0.01)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B, functionals as Fn
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
stop = False
def beat():
    t0 = time.time()
    while not stop:
        time.sleep(45); print(f"  ... {time.time() - t0:.0f}s", flush=True)
threading.Thread(target=beat, daemon=True).start()
nu = 0.001
for n in [int(a) for a in sys.argv[1:]] or [24]:
    t0 = time.time(); m = M.dfg_pillar_mesh(n, lattice=os.environ.get("DFG_LATTICE", "cubic")); m = M.reorder_for_locality(m)[0]; t1 = time.time()
    P = FlowProblem(m, B.dfg_bcs(m), reynolds=1.0 / nu, ksp_max_it=3000)
    U, r = P.stokes_solve()
    w, res = P.newton_solve(U.clone())
    torch.cuda.synchronize(); t2 = time.time()
    wh = w.cpu().numpy()
    f = Fn.boundary_traction_force(m, wh, nu, m.meta["tags"]["obstacle"])
    cd, cl = Fn.drag_lift_coefficients(f)
    # pressure difference front - back of the pillar at mid height (benchmark quantity Delta p ~ 0.165-0.175)
    W4 = wh.reshape(-1, 4)
    def p_at(x, y, z):
        i = np.argmin(((m.points - np.array([x, y, z])) ** 2).sum(axis=1)); return W4[i, 3]
    dp = p_at(0.45, 0.2, 0.205) - p_at(0.55, 0.2, 0.205)
    print(f"n={n} h={0.41/n:.4f}: {m.num_nodes} nodes {m.num_tets} tets (mesh {t1-t0:.0f}s) | stokes its {r.its} | newton its {res.its} "
          f"reason {res.reason} ksp {res.ksp_its} |F| {res.fnorms[-1]:.1e} | C_d {cd:.4f} C_l {cl:.5f} dp {dp:.4f} | solve {t2-t1:.1f}s", flush=True)
    P.close()
stop = True
