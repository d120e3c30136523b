"""round 5 diagnostic: is it the MESH TYPE of the body-fitted nozzle channel (a Delaunay cross-section extruded into prisms, 3 tets
each) that costs iterations, or its geometry?  The plain duct flow (no nozzle: no-slip walls, u = (1,0,0) on x = 0, p = 0 on x = 4,
Re 50; Dirichlet data set by coordinates, the same function for every mesh) on
  (a) the Kuhn duct lattice, (b) a hexagonal-lattice cross-section extruded over uniform planes at the same spacing,
  (c) the body-centred Delaunay channel of round 3 (config 4u's mesh type).
usage: python scripts/gpu_r5_prism_vs_kuhn.py [n_across = 20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import Delaunay
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.mesh import TetMesh
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
h = 1.0 / n

def bcs_by_coordinates(m):
    p = m.points
    nd = 4 * len(p)
    mask, g = np.zeros(nd, np.uint8), np.zeros(nd)
    wall = (np.abs(np.abs(p[:, 1]) - 0.5) < 1e-9) | (np.abs(np.abs(p[:, 2]) - 0.5) < 1e-9)
    inlet = (np.abs(p[:, 0]) < 1e-9) & ~wall
    outlet = np.abs(p[:, 0] - 4.0) < 1e-9
    for c in range(3):
        mask[4 * np.nonzero(wall | inlet)[0] + c] = 1
    g[4 * np.nonzero(inlet)[0]] = 1.0
    mask[4 * np.nonzero(outlet)[0] + 3] = 1
    return mask, g

def square_prism_duct(h, good_order=True):
    # cross-section: the square lattice, every cell cut along the same diagonal (a tiny shear decides qhull's tie), the prisms cut
    # along diagonals that follow the order (y ascending, x descending): right angle at the middle vertex = the Kuhn cells
    k = int(round(1.0 / h))
    t = np.linspace(-0.5, 0.5, k + 1)
    X, Y = np.meshgrid(t, t, indexing="xy")
    p2 = np.stack([X.ravel(), Y.ravel()], 1)
    q = p2.copy(); q[:, 0] += 1e-4 * p2[:, 1]
    tri = Delaunay(q).simplices
    a, bb, c = p2[tri[:, 0]], p2[tri[:, 1]], p2[tri[:, 2]]
    area2 = np.abs((bb[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (bb[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))
    tri = tri[area2 > 1e-3 * h * h]
    rank = np.empty(len(p2), np.int64)
    rank[np.lexsort((-p2[:, 0], p2[:, 1]) if good_order else (p2[:, 0], p2[:, 1]))] = np.arange(len(p2))
    return extrude(p2, tri, rank, h, "square-lattice prisms")

def extrude(p2, tri, rank, h, name):
    N = len(p2)
    xs = np.linspace(0.0, 4.0, int(round(4.0 / h)) + 1)
    pts = np.zeros((N * len(xs), 3))
    for q, x in enumerate(xs):
        pts[q * N:(q + 1) * N, 0] = x
        pts[q * N:(q + 1) * N, 1:] = p2
    o = np.argsort(rank[tri], axis=1)
    t3 = np.take_along_axis(tri.astype(np.int64), o, axis=1)
    tets = []
    for q in range(len(xs) - 1):
        v0, v1, v2 = (t3[:, j] + q * N for j in range(3))
        w0, w1, w2 = v0 + N, v1 + N, v2 + N
        tets += [np.stack([v0, v1, v2, w2], 1), np.stack([v0, v1, w2, w1], 1), np.stack([v0, w0, w1, w2], 1)]
    tets = np.concatenate(tets)
    P = pts
    d = np.einsum("ij,ij->i", np.cross(P[tets[:, 1]] - P[tets[:, 0]], P[tets[:, 2]] - P[tets[:, 0]]), P[tets[:, 3]] - P[tets[:, 0]])
    tets[d < 0] = tets[d < 0][:, [0, 2, 1, 3]]
    return TetMesh(pts, tets.astype(np.int32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32), name=name, meta={})

def max_dihedral(m, sample=20000):
    import itertools
    t = m.tets[:: max(1, len(m.tets) // sample)]
    P = m.points[t]
    worst = np.zeros(len(t))
    for (i, j) in itertools.combinations(range(4), 2):
        k, l = [q for q in range(4) if q not in (i, j)]
        e = P[:, j] - P[:, i]; e /= np.linalg.norm(e, axis=1)[:, None]
        a = P[:, k] - P[:, i]; a -= (a * e).sum(1)[:, None] * e
        b = P[:, l] - P[:, i]; b -= (b * e).sum(1)[:, None] * e
        ang = np.degrees(np.arccos(np.clip((a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1), -1, 1)))
        worst = np.maximum(worst, ang)
    return np.percentile(worst, [50, 90, 100]).round(1)

def hex_prism_duct(h):
    # cross-section: boundary points at spacing h, hexagonal interior lattice kept 0.6 h off the boundary
    k = int(round(1.0 / h))
    t = np.linspace(-0.5, 0.5, k + 1)
    b = np.concatenate([np.stack([t, np.full_like(t, -0.5)], 1), np.stack([t, np.full_like(t, 0.5)], 1),
                        np.stack([np.full_like(t[1:-1], -0.5), t[1:-1]], 1), np.stack([np.full_like(t[1:-1], 0.5), t[1:-1]], 1)])
    rows = []
    dy = h * np.sqrt(3.0) / 2.0
    ys = np.arange(-0.5 + 0.8 * h, 0.5 - 0.6 * h, dy)
    for i, y in enumerate(ys):
        xs = np.arange(-0.5 + 0.8 * h + (0.5 * h if i % 2 else 0.0), 0.5 - 0.6 * h, h)
        rows.append(np.stack([xs, np.full_like(xs, y)], 1))
    p2 = np.concatenate([b] + rows)
    tri = Delaunay(p2).simplices
    a, bb, c = p2[tri[:, 0]], p2[tri[:, 1]], p2[tri[:, 2]]
    area2 = np.abs((bb[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (bb[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))
    tri = tri[area2 > 1e-3 * h * h]
    N = len(p2)
    xs = np.linspace(0.0, 4.0, int(round(4.0 / h)) + 1)
    pts = np.zeros((N * len(xs), 3))
    for q, x in enumerate(xs):
        pts[q * N:(q + 1) * N, 0] = x
        pts[q * N:(q + 1) * N, 1:] = p2
    t3 = np.sort(tri.astype(np.int64), axis=1)
    tets = []
    for q in range(len(xs) - 1):
        v0, v1, v2 = (t3[:, j] + q * N for j in range(3))
        w0, w1, w2 = v0 + N, v1 + N, v2 + N
        tets += [np.stack([v0, v1, v2, w2], 1), np.stack([v0, v1, w2, w1], 1), np.stack([v0, w0, w1, w2], 1)]
    tets = np.concatenate(tets)
    # positive orientation
    P = pts
    d = np.einsum("ij,ij->i", np.cross(P[tets[:, 1]] - P[tets[:, 0]], P[tets[:, 2]] - P[tets[:, 0]]), P[tets[:, 3]] - P[tets[:, 0]])
    tets[d < 0] = tets[d < 0][:, [0, 2, 1, 3]]
    return TetMesh(pts, tets.astype(np.int32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32), name="hex-prism duct", meta={})

def run(name, m):
    P = FlowProblem(m, bcs_by_coordinates(m), reynolds=50.0)
    U, r = P.stokes_solve(); w, nn = P.newton_solve(U.clone())
    rows = [hh["rows"] for hh in P.hierarchy()]
    P.close()
    print(f"{name}: max dihedral per cell 50 / 90 / 100 %: {max_dihedral(m)};  {m.num_tets} tets, {m.num_nodes} nodes, stokes {r.its}, newton {nn.its} its {nn.ksp_its} ksp ({nn.ksp_its / nn.its:.1f}/step) reason {nn.reason}; "
          f"rows {rows} ratios {[round(rows[i] / rows[i + 1], 2) for i in range(len(rows) - 1)]}", flush=True)

run("(a) Kuhn lattice", M.duct_mesh((4 * n, n, n), 4.0))
run("(b) hexagonal cross-section, prisms", hex_prism_duct(h))
run("(c) body-centred Delaunay", M.delaunay_channel_mesh(n, lattice="bcc"))
run("(d) square-lattice cross-section, prisms cut in the order that keeps the right angle in the middle", square_prism_duct(h, True))
def renumbered(m, name):
    # node ids along x slowest, z, then y DESCENDING fastest: the cells' common diagonal then runs towards increasing ids, as on the
    # Kuhn lattice of (a); cells and their local vertex order are untouched
    p = m.points
    order = np.lexsort((-p[:, 1].round(9), p[:, 2].round(9), p[:, 0].round(9)))
    new = np.empty(len(p), np.int64); new[order] = np.arange(len(p))
    return TetMesh(p[order], new[m.tets].astype(np.int32), m.facets, m.facet_tags, name=name, meta={})
run("(f) = (d) with the nodes renumbered so that the cells' common diagonal runs towards increasing ids", renumbered(square_prism_duct(h, True), "renumbered"))
run("(g) = (b) with the nodes renumbered the same way", renumbered(hex_prism_duct(h), "renumbered hex"))
def strips_duct(h):
    # (h) the body-fitted nozzle channel's OWN mesh type without its flow: the cross-section that conforms to the image's two contours
    # (square lattice + strips of general triangles along them), uniform planes, the nozzle shrunk to the first cell layer; uniform
    # inflow through both inlets, no-slip on every wall facet
    from stabilized_navier_stokes_flow_fenicsx_amd import nozzle_mesh as NM
    from stabilized_navier_stokes_flow_fenicsx_amd.inlet_contours import solve_inlet_profiles
    img = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inlet_PlusF_final.png")
    data = solve_inlet_profiles(img, 0.5, max_pixels=1024)
    orig = NM.size_along_x
    NM.size_along_x = lambda x, lc_, x_extrude=0.5, growth=0.35, far=2.0: np.full_like(np.asarray(x, dtype=np.float64), h)
    try:
        m = NM.nozzle_channel_mesh(data.contour_inner[:, ::-1], data.contour_outer[:, ::-1], h / 0.75, x_extrude=h, cross_size=h,
                                   lattice_behind=None)             # (the contours' strips through the whole channel: the first versions)
    finally:
        NM.size_along_x = orig
    return m

def bcs_by_tags(m):
    t = m.meta["tags"]
    nd = 4 * m.num_nodes
    mask, g = np.zeros(nd, np.uint8), np.zeros(nd)
    wall = m.facet_nodes(t["wall"])
    inlet = np.setdiff1d(np.union1d(m.facet_nodes(t["inlet_1"]), m.facet_nodes(t["inlet_2"])), wall)
    for c in range(3):
        mask[4 * np.union1d(wall, inlet) + c] = 1
    g[4 * inlet] = 1.0
    mask[4 * m.facet_nodes(t["outlet"]) + 3] = 1
    return mask, g

def run_tags(name, m):
    P = FlowProblem(m, bcs_by_tags(m), reynolds=50.0)
    U, r = P.stokes_solve(); w, nn = P.newton_solve(U.clone())
    rows = [hh["rows"] for hh in P.hierarchy()]
    P.close()
    print(f"{name}: max dihedral per cell 50 / 90 / 100 %: {max_dihedral(m)};  {m.num_tets} tets, {m.num_nodes} nodes, stokes {r.its}, newton {nn.its} its {nn.ksp_its} ksp "
          f"({nn.ksp_its / nn.its:.1f}/step) reason {nn.reason}; rows {rows} ratios {[round(rows[i] / rows[i + 1], 2) for i in range(len(rows) - 1)]}", flush=True)

run_tags("(h) the nozzle channel's cross-section (square lattice + contour strips), uniform planes, nozzle one cell long, plain inflow", strips_duct(h))
run("(e) the same cross-section, prisms cut in plain (y, x) order", square_prism_duct(h, False))
