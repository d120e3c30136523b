"""round 2: does the fp16 row-scaled preconditioner copy (amg_f32_matrix = 2) cost iterations anywhere?
fp16 vs fp32 vs fp64 smoother matrices on harder cases than the headline duct: unstructured Delaunay meshes (DFG-3D pillar,
jittered duct with slivers), higher Reynolds numbers, the lid-driven cavity."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cases = [
    ("DFG-3D pillar bcc n=24 (Re 20)", lambda: (lambda m: (m, B.dfg_bcs(m)))(M.reorder_for_locality(M.dfg_pillar_mesh(24, lattice="bcc"))[0]), 1000.0),
    ("Delaunay duct n=20 (slivers), Re 100", lambda: (lambda m: (m, B.duct_bcs(m)))(M.delaunay_duct_mesh(20, 2.0, seed=1)), 100.0),
    ("cavity 40^3 Re 400", lambda: (lambda m: (m, B.cavity_bcs(m)))(M.cavity_mesh(40)), 400.0),
    ("duct 200x50x50 Re 500", lambda: (lambda m: (m, B.duct_bcs(m)))(M.duct_mesh((200, 50, 50), 4.0)), 500.0),
    ("jittered duct 120x30x30 Re 200", lambda: (lambda m: (m, B.duct_bcs(m)))(M.duct_mesh((120, 30, 30), 4.0, jitter=0.2)), 200.0),
]
for name, make, Re in cases:
    m, bcs = make()
    row = []
    ref = None
    for fmt in (0, 1, 2):
        P = FlowProblem(m, bcs, reynolds=Re, amg_f32_matrix=fmt, ksp_max_it=3000)
        U, r = P.stokes_solve()
        t = time.time(); w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); dt = time.time() - t
        wh = w.cpu().numpy()
        if ref is None:
            ref = wh
        row.append(f"fmt {fmt}: stokes {r.its} newton {n.its}/{n.reason} ksp {n.ksp_its} {dt:.2f}s diff {np.linalg.norm(wh - ref) / np.linalg.norm(ref):.1e}")
        P.close()
    print(f"{name} ({m.num_tets} tets): " + " | ".join(row), flush=True)
