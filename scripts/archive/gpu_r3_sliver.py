"""round 3: where do sliver-rich meshes lose their convergence?  Per-level damping of the automatic estimate on the body-centred
(near-regular) and the jittered-cubic (sliver-rich) Delaunay channel, and the effect of a hand-set damping."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, m in (("bcc", M.delaunay_channel_mesh(n, lattice="bcc")), ("cubic", M.delaunay_channel_mesh(int(round(n * 2 ** (1 / 3))), lattice="cubic"))):
    X = m.points[m.tets]
    vol = np.abs(np.linalg.det(np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2))) / 6
    e = np.stack([np.linalg.norm(X[:, a] - X[:, b], axis=1) for a in range(4) for b in range(a + 1, 4)], axis=1)
    q = 6 * np.sqrt(2) * vol / (np.sqrt((e ** 2).mean(axis=1)) ** 3)          # 1 for a regular tet
    print(f"== {name}: {m.num_tets} tets, quality quantiles 0.1 % {np.quantile(q, 0.001):.4f} 1 % {np.quantile(q, 0.01):.3f} median {np.median(q):.2f}", flush=True)
    for kw in (dict(), dict(amg_omega=0.5), dict(amg_omega=0.35)):
        P = FlowProblem(m, B.channel_bcs(m, *B.two_stream_profiles(0.5)), reynolds=50.0, monitor=0, **kw)
        P.set_options(monitor=1)
        U, r = P.stokes_solve()
        P.set_options(monitor=0)
        w, nr = P.newton_solve(U.clone())
        print(f"OPTS {kw}: stokes its {r.its}, newton {nr.its} its reason {nr.reason}, ksp its/step {nr.ksp_its / max(1, nr.its):.1f}", flush=True)
        P.close()
