"""2-D triangle meshes, Dirichlet data and the drag / lift functional of the reference's 2-D scripts.

  * ``rectangle_mesh``   dolfinx ``create_rectangle(..., CellType.triangle)`` (default diagonal "right") as used by
                         LidDrivenFlow/LidDrivenNavierStokesFlow.py:29-30
  * ``dfg_2d_mesh``      the DFG 2D-1 channel of Validation_Flow/dfg_pillar_2D.geo (2.2 x 0.41, cylinder r = 0.05 at
                         (0.2, 0.2), physical groups inlet 2 / outlet 3 / walls 4 / obstacle 5, DFG_2D_Validation.py:
                         60-64) meshed WITHOUT gmsh: graded rings round the cylinder + a hexagonal background lattice,
                         triangulated by scipy's Delaunay
  * ``read_msh_2d``      triangles + tagged boundary lines from a gmsh ASCII file (DFG_2D_Validation.py:28)
  * ``cavity2d_bcs`` / ``dfg2d_bcs``   the scripts' ``dirichletbc`` lists (last entry wins on shared dofs)
  * ``drag_lift_2d``     DFG_2D_Validation.py:195-200

The solver keeps 4 dofs per node [ux, uy, uz, p] for 2-D problems too (uz constrained to 0 inside libsns.so), so
Dirichlet masks / values and state vectors have 4*num_nodes entries here as well.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .bcs import DirichletBC, DirichletSet

DFG2D_TAGS = {"inlet": 2, "outlet": 3, "walls": 4, "obstacle": 5}      # DFG_2D_Validation.py:60-64
CAVITY2D_TAGS = {"noslip": 1, "lid": 2}


@dataclass
class TriMesh:
    points: np.ndarray          # (N,2) float64
    tris: np.ndarray            # (E,3) int32
    facets: np.ndarray          # (F,2) int32 boundary edges
    facet_tags: np.ndarray      # (F,) int32
    name: str = "mesh2d"
    meta: dict = field(default_factory=dict)
    dim: int = 2

    @property
    def num_nodes(self) -> int:
        return int(self.points.shape[0])

    @property
    def num_cells(self) -> int:
        return int(self.tris.shape[0])

    @property
    def num_dofs(self) -> int:
        return 4 * self.num_nodes

    def find(self, tag: int) -> np.ndarray:
        return np.nonzero(self.facet_tags == tag)[0]

    def facet_nodes(self, tag: int) -> np.ndarray:
        return np.unique(self.facets[self.facet_tags == tag].ravel())


def boundary_edges(tris: np.ndarray) -> np.ndarray:
    """Edges that belong to exactly one triangle."""
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]]).astype(np.int64)
    es = np.sort(e, axis=1)
    n = int(tris.max()) + 1
    key = es[:, 0] * n + es[:, 1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.ones(ks.size, dtype=bool)
    first[1:] = ks[1:] != ks[:-1]
    last = np.ones(ks.size, dtype=bool)
    last[:-1] = ks[1:] != ks[:-1]
    return e[order[first & last]].astype(np.int32)


def rectangle_mesh(nx: int, ny: int | None = None, lo=(0.0, 0.0), hi=(1.0, 1.0), diagonal: str = "right") -> TriMesh:
    """``create_rectangle(comm, [lo, hi], [nx, ny], CellType.triangle)``: vertex (ix, iy) has id iy*(nx+1)+ix, every
    cell is split along the diagonal v0-v3 ("right") into (v0, v1, v3), (v0, v2, v3).  Edges on y = hi are tagged
    ``lid``, the rest ``noslip`` (the two marker functions of LidDrivenNavierStokesFlow.py:33-40)."""
    ny = nx if ny is None else ny
    xs = np.linspace(lo[0], hi[0], nx + 1)
    ys = np.linspace(lo[1], hi[1], ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")                 # row iy, column ix
    pts = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (iy * (nx + 1) + ix).ravel()
    v1, v2, v3 = v0 + 1, v0 + nx + 1, v0 + nx + 2
    if diagonal == "right":
        tris = np.stack([np.stack([v0, v1, v3], 1), np.stack([v0, v2, v3], 1)], axis=1).reshape(-1, 3)
    elif diagonal == "left":
        tris = np.stack([np.stack([v0, v1, v2], 1), np.stack([v1, v2, v3], 1)], axis=1).reshape(-1, 3)
    else:
        raise ValueError("diagonal must be 'right' or 'left'")
    tris = tris.astype(np.int32)
    fac = boundary_edges(tris)
    mid = pts[fac].mean(axis=1)
    tags = np.full(len(fac), CAVITY2D_TAGS["noslip"], dtype=np.int32)
    tags[np.isclose(mid[:, 1], hi[1])] = CAVITY2D_TAGS["lid"]
    return TriMesh(pts, tris, fac, tags, name="rectangle",
                   meta={"cells": (nx, ny), "lo": tuple(lo), "hi": tuple(hi), "tags": dict(CAVITY2D_TAGS), "kind": "cavity2d"})


def _morton2(points: np.ndarray, bits: int = 24) -> np.ndarray:
    lo, hi = points.min(axis=0), points.max(axis=0)
    ext = float(np.max(hi - lo))
    q = ((points - lo) / ext * ((1 << bits) - 1)).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0xFFFFFFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


def reorder_for_locality_2d(mesh: TriMesh) -> TriMesh:
    """Nodes along a Morton curve, triangles by their lowest node (SpMV / assembly gather locality)."""
    perm = np.argsort(_morton2(mesh.points), kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    tris = inv[mesh.tris].astype(np.int32)
    order = np.argsort(tris.min(axis=1), kind="stable")
    fac = inv[mesh.facets].astype(np.int32) if len(mesh.facets) else mesh.facets
    return TriMesh(np.ascontiguousarray(mesh.points[perm]), np.ascontiguousarray(tris[order]), fac,
                   mesh.facet_tags.copy(), name=mesh.name, meta=dict(mesh.meta))


def dfg_2d_mesh(n: float = 1.0, *, length: float = 2.2, width: float = 0.41, cx: float = 0.2, cy: float = 0.2,
                radius: float = 0.05, h_bg: float | None = None, h_cyl: float | None = None, growth: float = 0.12,
                smooth: int = 3) -> TriMesh:
    """DFG 2D-1 geometry (dfg_pillar_2D.geo:4-9), graded like the .geo's size fields but without gmsh.

    Level ``n`` scales every size by 1/n: background h_bg = 0.02/n, on the cylinder h_cyl = 0.002/n, growing by
    ``growth`` per ring.  The cylinder is the inscribed polygon through the first ring (its vertices lie ON the
    circle).  A few Laplacian smoothing passes (interior nodes only, re-triangulated afterwards) even out the
    junction between the rings and the lattice."""
    from scipy.spatial import Delaunay
    h_bg = 0.02 / n if h_bg is None else h_bg
    h_cyl = 0.002 / n if h_cyl is None else h_cyl
    c = np.array([cx, cy])
    r_max = min(cx, cy, width - cy) - 1.5 * h_bg                 # rings stay clear of the channel walls
    rings, r, hk, k = [], radius, h_cyl, 0
    while True:
        nk = max(8, int(round(2 * np.pi * r / hk)))
        ang = (np.arange(nk) + 0.5 * (k % 2)) * (2 * np.pi / nk)
        rings.append(c + r * np.stack([np.cos(ang), np.sin(ang)], axis=1))
        step = 2 * np.pi * r / nk
        r_next = r + step * (np.sqrt(3.0) / 2.0)
        if step >= h_bg * 0.999 or r_next > r_max:
            break
        r, hk, k = r_next, min(h_bg, step * (1.0 + growth)), k + 1
    r_last = r
    n_obst = len(rings[0])
    # hexagonal background lattice fitted to the box
    ny = max(2, int(round(width / (h_bg * np.sqrt(3.0) / 2.0))))
    nx = max(2, int(round(length / h_bg)))
    dx, dy = length / nx, width / ny
    bg = []
    for j in range(ny + 1):
        y = j * dy
        if j % 2 == 0:
            x = np.arange(nx + 1) * dx
        else:
            x = np.concatenate([[0.0], (np.arange(nx) + 0.5) * dx, [length]])
        bg.append(np.stack([x, np.full_like(x, y)], axis=1))
    bg = np.concatenate(bg)
    on_box = np.isclose(bg[:, 0], 0) | np.isclose(bg[:, 0], length) | np.isclose(bg[:, 1], 0) | np.isclose(bg[:, 1], width)
    d = np.linalg.norm(bg - c, axis=1)
    bg_keep = d > r_last + 0.75 * max(dx, 2 * np.pi * r_last / len(rings[-1]))
    bg, on_box = bg[bg_keep], on_box[bg_keep]
    pts = np.concatenate(rings + [bg])
    n_ring = sum(len(q) for q in rings)
    fixed = np.zeros(len(pts), dtype=bool)
    fixed[:n_obst] = True
    fixed[n_ring:] = on_box
    is_obst = np.zeros(len(pts), dtype=bool)
    is_obst[:n_obst] = True

    def triangulate(p):
        t = Delaunay(p).simplices.astype(np.int64)
        t = t[~np.all(is_obst[t], axis=1)]                       # triangles inside the cylinder polygon
        # drop degenerate slivers along the straight box sides (collinear boundary points)
        a = p[t]
        area2 = np.abs((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
        return t[area2 > 1e-14 * h_bg * h_bg]

    tris = triangulate(pts)
    for _ in range(max(0, smooth)):
        e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
        e = np.concatenate([e, e[:, ::-1]])
        acc = np.zeros_like(pts)
        cnt = np.zeros(len(pts))
        np.add.at(acc, e[:, 0], pts[e[:, 1]])
        np.add.at(cnt, e[:, 0], 1.0)
        new = acc / np.maximum(cnt, 1.0)[:, None]
        pts = np.where(fixed[:, None], pts, 0.5 * pts + 0.5 * new)
        tris = triangulate(pts)
    # orient counter-clockwise
    a = pts[tris]
    cw = ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1])) < 0
    tris[cw] = tris[cw][:, [0, 2, 1]]
    used = np.zeros(len(pts), dtype=bool)
    used[tris.ravel()] = True
    if not used.all():
        new_id = -np.ones(len(pts), dtype=np.int64)
        new_id[used] = np.arange(int(used.sum()))
        pts, tris = pts[used], new_id[tris]
    tris = tris.astype(np.int32)
    fac = boundary_edges(tris)
    mid = pts[fac].mean(axis=1)
    tol = 1e-9
    tags = np.full(len(fac), DFG2D_TAGS["obstacle"], dtype=np.int32)
    tags[np.abs(mid[:, 1]) < tol] = DFG2D_TAGS["walls"]
    tags[np.abs(mid[:, 1] - width) < tol] = DFG2D_TAGS["walls"]
    tags[np.abs(mid[:, 0]) < tol] = DFG2D_TAGS["inlet"]
    tags[np.abs(mid[:, 0] - length) < tol] = DFG2D_TAGS["outlet"]
    m = TriMesh(pts, tris, fac, tags, name="dfg2d",
                meta={"kind": "dfg2d", "tags": dict(DFG2D_TAGS), "length": length, "width": width, "centre": (cx, cy),
                      "radius": radius, "h_bg": h_bg, "h_cyl": h_cyl, "n_obstacle_edges": n_obst, "level": n})
    return reorder_for_locality_2d(m)


def triangle_quality(mesh: TriMesh) -> np.ndarray:
    """4 sqrt(3) area / (sum of squared edge lengths): 1 for equilateral triangles."""
    a = mesh.points[mesh.tris]
    e = [a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 0] - a[:, 2]]
    area = 0.5 * np.abs(e[0][:, 0] * (-e[2][:, 1]) - e[0][:, 1] * (-e[2][:, 0]))
    return 4 * np.sqrt(3.0) * area / sum((v * v).sum(axis=1) for v in e)


def read_msh_2d(path: str, reorder: bool = True) -> TriMesh:
    """Triangles (gmsh type 2) + boundary lines (type 1) carrying their physical-group id, from an ASCII .msh 2.2 or
    4.1 file -- what ``gmshio.read_from_msh(file, comm, 0, gdim=2)`` hands DFG_2D_Validation.py:28."""
    with open(path, "r") as fh:
        lines = fh.read().split("\n")
    sec, i = {}, 0
    while i < len(lines):
        ln = lines[i].strip()
        if ln.startswith("$") and not ln.startswith("$End"):
            nm, j = ln[1:], i + 1
            while lines[j].strip() != "$End" + nm:
                j += 1
            sec[nm] = lines[i + 1:j]
            i = j
        i += 1
    head = sec["MeshFormat"][0].split()
    if int(head[1]) != 0:
        raise ValueError("binary .msh files are not supported; write ASCII")
    tris, edges, etags = [], [], []
    if float(head[0]) < 3.0:
        nl = sec["Nodes"]
        arr = np.array([ln.split() for ln in nl[1:1 + int(nl[0])]], dtype=np.float64)
        ids, pts = arr[:, 0].astype(np.int64), arr[:, 1:3]
        el = sec["Elements"]
        for ln in el[1:1 + int(el[0])]:
            t = ln.split()
            et, ntag = int(t[1]), int(t[2])
            phys = int(t[3]) if ntag > 0 else 0
            nod = [int(v) for v in t[3 + ntag:]]
            if et == 1 and phys > 0:
                edges.append(nod); etags.append(phys)
            elif et == 2:
                tris.append(nod)
    else:
        ent_phys = {}
        el = sec["Entities"]
        npnt, ncur, nsur, _ = (int(v) for v in el[0].split())
        k = 1 + npnt
        for ln in el[k:k + ncur]:
            t = ln.split()
            if int(t[7]) > 0:
                ent_phys[int(t[0])] = abs(int(t[8]))
        nl = sec["Nodes"]
        nblocks, nnodes = int(nl[0].split()[0]), int(nl[0].split()[1])
        ids = np.empty(nnodes, dtype=np.int64)
        pts = np.empty((nnodes, 2))
        k, w = 1, 0
        for _ in range(nblocks):
            cnt = int(nl[k].split()[3])
            ids[w:w + cnt] = [int(v) for v in nl[k + 1:k + 1 + cnt]]
            pts[w:w + cnt] = [[float(v) for v in ln.split()[:2]] for ln in nl[k + 1 + cnt:k + 1 + 2 * cnt]]
            k += 1 + 2 * cnt
            w += cnt
        el = sec["Elements"]
        k = 1
        for _ in range(int(el[0].split()[0])):
            _dim, etag, et, cnt = (int(v) for v in el[k].split())
            rows = [[int(v) for v in ln.split()[1:]] for ln in el[k + 1:k + 1 + cnt]]
            if et == 1 and etag in ent_phys:
                edges += rows; etags += [ent_phys[etag]] * cnt
            elif et == 2:
                tris += rows
            k += 1 + cnt
    remap = -np.ones(int(ids.max()) + 1, dtype=np.int64)
    remap[ids] = np.arange(ids.size)
    tris = remap[np.array(tris, dtype=np.int64).reshape(-1, 3)]
    edges = remap[np.array(edges, dtype=np.int64).reshape(-1, 2)]
    used = np.zeros(ids.size, dtype=bool)
    used[tris.ravel()] = True
    new = -np.ones(ids.size, dtype=np.int64)
    new[used] = np.arange(int(used.sum()))
    m = TriMesh(np.ascontiguousarray(pts[used]), new[tris].astype(np.int32), new[edges].astype(np.int32),
                np.asarray(etags, dtype=np.int32), name=path, meta={"kind": "msh2d", "tags": dict(DFG2D_TAGS)})
    return reorder_for_locality_2d(m) if reorder else m


def write_msh2_2d(mesh: TriMesh, path: str) -> None:
    """ASCII gmsh-2.2 file with the boundary lines in their physical groups (tests, hand-over to the reference)."""
    with open(path, "w") as fh:
        fh.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % mesh.num_nodes)
        for i, p in enumerate(mesh.points):
            fh.write("%d %.17g %.17g 0\n" % (i + 1, p[0], p[1]))
        fh.write("$EndNodes\n$Elements\n%d\n" % (len(mesh.facets) + mesh.num_cells))
        e = 1
        for f, t in zip(mesh.facets, mesh.facet_tags):
            fh.write("%d 1 2 %d %d %d %d\n" % (e, t, t, f[0] + 1, f[1] + 1))
            e += 1
        for c in mesh.tris:
            fh.write("%d 2 2 1 1 %d %d %d\n" % (e, c[0] + 1, c[1] + 1, c[2] + 1))
            e += 1
        fh.write("$EndElements\n")


# ---- Dirichlet data -----------------------------------------------------------------------------------------
def _vel2(nodes, fn_or_val, pts) -> DirichletBC:
    nodes = np.asarray(nodes, dtype=np.int64)
    if callable(fn_or_val):
        vals = np.asarray(fn_or_val(pts[nodes]), dtype=np.float64).reshape(len(nodes), 2)
    else:
        vals = np.broadcast_to(np.asarray(fn_or_val, dtype=np.float64), (len(nodes), 2)).copy()
    return DirichletBC(nodes, (0, 1), vals)


def cavity2d_bcs(mesh: TriMesh, lid_velocity=(1.0, 0.0)) -> DirichletSet:
    """bcs = [noslip (x=0, x=1, y=0), lid u=(1,0) on y=1, p=0 at the origin] (LidDrivenNavierStokesFlow.py:57-77);
    the lid entry comes later in the list, so the two top corners move with the lid."""
    t = mesh.meta["tags"]
    origin = np.nonzero(np.all(np.isclose(mesh.points, 0.0), axis=1))[0]
    return DirichletSet(mesh, [
        _vel2(mesh.facet_nodes(t["noslip"]), (0.0, 0.0), mesh.points),
        _vel2(mesh.facet_nodes(t["lid"]), lid_velocity, mesh.points),
        DirichletBC(origin.astype(np.int64), (3,), np.zeros((len(origin), 1))),
    ])


def dfg2d_bcs(mesh: TriMesh, u_max: float = 0.3) -> DirichletSet:
    """bc = [inflow, walls, obstacle] (DFG_2D_Validation.py:50-90): u_x = 4 u_max y (H - y) / H^2 at the inlet (:52),
    no slip on walls and obstacle; the outlet pressure condition the script builds (:82-85) is NOT in the list it
    passes on (:90), so the outlet is the natural boundary."""
    t = mesh.meta["tags"]
    H = 0.41

    def inflow(x):
        return np.stack([4 * x[:, 1] * u_max * (H - x[:, 1]) / H ** 2, np.zeros(len(x))], axis=1)

    return DirichletSet(mesh, [
        _vel2(mesh.facet_nodes(t["inlet"]), inflow, mesh.points),
        _vel2(mesh.facet_nodes(t["walls"]), (0.0, 0.0), mesh.points),
        _vel2(mesh.facet_nodes(t["obstacle"]), (0.0, 0.0), mesh.points),
    ])


def dfg2d_slab_problem(n: float, u_max: float = 0.3):
    """The DFG 2D-1 problem posed on the 3-D tet path: the built-in triangulation extruded to a one-cell slab
    (``mesh.extrude_tri_mesh``, thickness = the mean triangle size) with u_z = 0 on both z planes -- every node lies on
    one, so the 3-D problem is z-independent in the continuum (the prism split is not, which costs a 0.3 % difference
    between the two planes on the coarse levels) and its drag / lift per unit depth must converge to the same constants (DFG_2D_Validation.py:202-203).  Inlet profile, no-slip sets and the natural outlet as ``dfg2d_bcs``.
    Returns (TetMesh, (mask, g), thickness)."""
    from . import mesh as M3
    m2 = dfg_2d_mesh(n)
    e = m2.points[m2.tris]
    a, b = e[:, 1] - e[:, 0], e[:, 2] - e[:, 0]
    thick = float(np.sqrt(np.abs(a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]).mean()))
    m3 = M3.extrude_tri_mesh(m2.points, m2.tris, m2.facets, m2.facet_tags, thick, tags=m2.meta["tags"])
    t = m3.meta["tags"]
    N = m3.num_nodes
    mask = np.zeros(4 * N, np.uint8)
    g = np.zeros(4 * N)
    mask[2::4] = 1
    H = 0.41
    for name in ("walls", "obstacle", "inlet"):
        nd = m3.facet_nodes(t[name])
        for c in range(3):
            mask[4 * nd + c] = 1
        if name == "inlet":
            y = m3.points[nd, 1]
            g[4 * nd] = 4 * u_max * y * (H - y) / H ** 2
    return m3, (mask, g), thick


# ---- drag / lift (DFG_2D_Validation.py:195-200) ----------------------------------------------------------------
DFG2D_CD_REF = 5.57953523384          # DFG_2D_Validation.py:203
DFG2D_CL_REF = 0.010618948146         # DFG_2D_Validation.py:202


def edge_parent_tris(mesh: TriMesh, edge_ids: np.ndarray) -> np.ndarray:
    f = np.sort(mesh.facets[edge_ids].astype(np.int64), axis=1)
    n = mesh.num_nodes
    want = f[:, 0] * n + f[:, 1]
    t = mesh.tris.astype(np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    e.sort(axis=1)
    key = e[:, 0] * n + e[:, 1]
    order = np.argsort(key, kind="stable")
    pos = np.searchsorted(key[order], want)
    if np.any(pos >= len(key)) or np.any(key[order][np.minimum(pos, len(key) - 1)] != want):
        raise ValueError("a boundary edge is not an edge of any triangle")
    return (order[pos] % len(t)).astype(np.int64)


def drag_lift_2d(mesh: TriMesh, w, nu: float, tag: int | None = None, U_mean: float = 0.2, L: float = 0.1):
    """(C_D, C_L) with n = -FacetNormal, u_t = (n_y, -n_x).u (:195-197):
       C_D =  2/(U^2 L) int_obstacle nu (grad(u_t).n) n_y - p n_x ds            (:198, written 2 / (0.1 * 0.2**2))
       C_L = -2/(U^2 L) int_obstacle nu (grad(u_t).n) n_x + p n_y ds            (:199)
    Exact for P1: grad u is constant in the triangle behind an edge and p is linear along it."""
    tag = DFG2D_TAGS["obstacle"] if tag is None else tag
    ids = mesh.find(tag)
    W = np.asarray(w, dtype=np.float64).reshape(-1, 4)
    par = edge_parent_tris(mesh, ids)
    tn = mesh.tris[par].astype(np.int64)
    X = mesh.points[tn]                                               # (F,3,2)
    J = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]], axis=2)
    K = np.linalg.inv(J)
    g = np.concatenate([-K.sum(axis=1, keepdims=True), K], axis=1)    # (F,3,2)
    gu = np.einsum("fai,faj->fij", W[tn][:, :, :2], g)
    fn = mesh.facets[ids].astype(np.int64)
    P = mesh.points[fn]
    tv = P[:, 1] - P[:, 0]
    ln = np.linalg.norm(tv, axis=1)
    nf = np.stack([tv[:, 1], -tv[:, 0]], axis=1) / ln[:, None]
    opp = tn.sum(axis=1) - fn.sum(axis=1)
    sgn = np.sign(np.einsum("fi,fi->f", nf, P[:, 0] - mesh.points[opp]))
    n = -nf * sgn[:, None]                                            # -FacetNormal
    tt = np.stack([n[:, 1], -n[:, 0]], axis=1)
    dut = np.einsum("fi,fij,fj->f", tt, gu, n)
    pm = W[fn][:, :, 3].mean(axis=1)
    c = 2.0 / (U_mean ** 2 * L)
    cd = c * float(np.sum(ln * (nu * dut * n[:, 1] - pm * n[:, 0])))
    cl = -c * float(np.sum(ln * (nu * dut * n[:, 0] + pm * n[:, 1])))
    return cd, cl
