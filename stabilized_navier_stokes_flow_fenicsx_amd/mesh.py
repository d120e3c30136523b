"""Tetrahedral meshes with tagged boundary facets (host side, numpy).

Plays the role of ``gmshio.model_to_mesh`` in the reference
(NavierStokes/NavierStokesChannelFlow.py:111, StokesFlow/DuctStokesFlow.py:144):
it hands the solver a tet connectivity whose *cell-local vertex order is kept
exactly as given* (the metric tensor G of the stabilisation depends on which
vertex is local vertex 0, NavierStokesChannelFlow.py:232-235) and a list of
boundary triangles carrying integer physical tags.

gmsh is not available offline, so besides a ``.msh`` (2.2 / 4.1 ASCII) reader
this module has a deterministic structured box mesher (6 Kuhn tets per hex
cell) that reproduces the reference geometries: the square duct
``[0,L]x[-.5,.5]^2`` of DuctStokesFlow.py:36-124 (tags inlet=3, outlet=4,
wall=5, :117), the 4x1x1 channel of image2gmsh3D.py:435-438 (tags inlet_1=1,
inlet_2=2, outlet=3, wall=4) and the unit-cube cavity that extends
LidDrivenNavierStokesFlow.py:29-43 to 3-D.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from itertools import permutations

import numpy as np

# physical tags of the reference scripts
DUCT_TAGS = {"inlet": 3, "outlet": 4, "wall": 5}            # DuctStokesFlow.py:117
CHANNEL_TAGS = {"inlet_1": 1, "inlet_2": 2, "outlet": 3, "wall": 4}  # image2gmsh3D.py:435-438
CAVITY_TAGS = {"lid": 1, "wall": 2}


@dataclass
class TetMesh:
    points: np.ndarray          # (N,3) float64
    tets: np.ndarray            # (E,4) int32, cell-local order preserved
    facets: np.ndarray          # (F,3) int32 boundary triangles
    facet_tags: np.ndarray      # (F,) int32
    name: str = "mesh"
    meta: dict = field(default_factory=dict)

    @property
    def num_nodes(self) -> int:
        return int(self.points.shape[0])

    @property
    def num_tets(self) -> int:
        return int(self.tets.shape[0])

    @property
    def num_dofs(self) -> int:
        return 4 * self.num_nodes

    def find(self, tag: int) -> np.ndarray:
        """Facet indices carrying ``tag`` (mirror of dolfinx ``MeshTags.find``)."""
        return np.nonzero(self.facet_tags == tag)[0]

    def facet_nodes(self, tag: int) -> np.ndarray:
        """Sorted unique vertex ids on facets tagged ``tag`` (the P1 analogue of
        ``locate_dofs_topological(..., ft.find(tag))``)."""
        return np.unique(self.facets[self.facet_tags == tag].ravel())


# --------------------------------------------------------------------------- #
# triangle mesh -> one-cell-thick slab of tets (quasi-2-D problems on the 3-D path)
# --------------------------------------------------------------------------- #
def extrude_tri_mesh(points2d: np.ndarray, tris: np.ndarray, edges: np.ndarray, edge_tags: np.ndarray,
                     thickness: float, *, zmin_tag: int = 6, zmax_tag: int = 7, tags: dict | None = None) -> TetMesh:
    """Slab z in [0, thickness] over a conforming triangulation: every triangle becomes a prism cut into 3 tets.  The
    vertical quad over an edge (a, b), a < b, is always cut along bottom-a -> top-b, so neighbouring prisms agree.
    Nodes: bottom plane 0..N-1, top plane N..2N-1.  Boundary facets: the extruded boundary edges keep their 2-D tag,
    the two z planes get ``zmin_tag`` / ``zmax_tag``.  With u_z = 0 imposed on both planes (every node lies on one)
    the continuous 3-D problem is the 2-D one, which is how the 3-D forms are checked against 2-D reference values."""
    p2 = np.asarray(points2d, dtype=np.float64)
    N = len(p2)
    pts = np.zeros((2 * N, 3))
    pts[:N, :2] = p2
    pts[N:, :2] = p2
    pts[N:, 2] = thickness
    t = np.sort(np.asarray(tris, dtype=np.int64), axis=1)              # v0 < v1 < v2
    v0, v1, v2 = t[:, 0], t[:, 1], t[:, 2]
    w0, w1, w2 = v0 + N, v1 + N, v2 + N
    tets = np.concatenate([np.stack([v0, v1, v2, w2], axis=1), np.stack([v0, v1, w2, w1], axis=1),
                           np.stack([v0, w0, w1, w2], axis=1)]).astype(np.int32)
    e = np.sort(np.asarray(edges, dtype=np.int64), axis=1)             # a < b
    a, b = e[:, 0], e[:, 1]
    side = np.concatenate([np.stack([a, b, b + N], axis=1), np.stack([a, b + N, a + N], axis=1)])
    side_tags = np.concatenate([edge_tags, edge_tags])
    tr = np.asarray(tris, dtype=np.int64)
    facets = np.concatenate([side, tr, tr + N]).astype(np.int32)
    ftags = np.concatenate([side_tags, np.full(len(tr), zmin_tag), np.full(len(tr), zmax_tag)]).astype(np.int32)
    meta = {"kind": "extruded", "thickness": float(thickness), "n2d": N,
            "tags": dict(tags or {}, zmin=zmin_tag, zmax=zmax_tag)}
    return TetMesh(pts, tets, facets, ftags, name="extruded", meta=meta)


# --------------------------------------------------------------------------- #
# structured box -> 6 Kuhn tets per cell
# --------------------------------------------------------------------------- #
_KUHN = []
for _perm in permutations(range(3)):
    _v = [np.zeros(3, dtype=np.int64)]
    for _ax in _perm:
        _n = _v[-1].copy()
        _n[_ax] += 1
        _v.append(_n)
    _KUHN.append(np.stack(_v))        # (4,3) corner offsets, v0=(0,0,0) ... v3=(1,1,1)
_KUHN = np.stack(_KUHN)               # (6,4,3)


def box_tet_mesh(lo, hi, cells, *, jitter: float = 0.0, seed: int = 1234,
                 name: str = "box", x_window=None) -> tuple[np.ndarray, np.ndarray, tuple]:
    """Nodes and Kuhn-6 tets of the box ``lo..hi`` with ``cells=(nx,ny,nz)``.

    Node id = (i*(ny+1) + j)*(nz+1) + k  (x slowest), so that slabs in x are
    contiguous id ranges (used by the multi-GPU slab partition) and the
    neighbours of a node live in three adjacent yz-planes (SpMV locality).
    Optional interior-node jitter (+-jitter*h) breaks structured
    cache luck; boundary nodes never move.
    ``x_window=(c0, c1)`` returns only the cell planes c0 <= i < c1 (node planes c0..c1, coordinates
    bit-identical to the full box; window node id = global id - c0*(ny+1)*(nz+1)): one rank's piece of
    a slab-partitioned run is meshed without ever building the global mesh.
    """
    nx, ny, nz = (int(c) for c in cells)
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    xs = np.linspace(lo[0], hi[0], nx + 1)
    if x_window is not None:
        c0, c1 = int(x_window[0]), int(x_window[1])
        if not (0 <= c0 < c1 <= nx):
            raise ValueError(f"x_window {x_window} outside 0..{nx}")
        if jitter > 0.0:
            raise ValueError("jitter is not supported on a window of the box")
        xs = xs[c0:c1 + 1]
        nx = c1 - c0
    ys = np.linspace(lo[1], hi[1], ny + 1)
    zs = np.linspace(lo[2], hi[2], nz + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        h = (hi - lo) / np.array([nx, ny, nz])
        I, J, K = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
        interior = ((I > 0) & (I < nx) & (J > 0) & (J < ny) & (K > 0) & (K < nz)).ravel()
        d = rng.uniform(-jitter, jitter, size=pts.shape) * h
        pts[interior] += d[interior]

    sy, sx = (nz + 1), (ny + 1) * (nz + 1)
    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = (ci * sx + cj * sy + ck).ravel().astype(np.int64)     # node id of cell corner (0,0,0)
    off = _KUHN[..., 0] * sx + _KUHN[..., 1] * sy + _KUHN[..., 2]  # (6,4)
    tets = (base[:, None, None] + off[None]).reshape(-1, 4).astype(np.int32)
    return pts, tets, (nx, ny, nz)


def _boundary_facets(tets: np.ndarray) -> np.ndarray:
    """Triangles that belong to exactly one tet (vertex order as in the tet)."""
    f = np.concatenate([tets[:, [1, 2, 3]], tets[:, [0, 2, 3]], tets[:, [0, 1, 3]], tets[:, [0, 1, 2]]])
    fs = np.sort(f.astype(np.int64), axis=1)
    n = int(tets.max()) + 1
    key = (fs[:, 0] * n + fs[:, 1]) * n + fs[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.ones(ks.size, dtype=bool)
    first[1:] = ks[1:] != ks[:-1]
    last = np.ones(ks.size, dtype=bool)
    last[:-1] = ks[1:] != ks[:-1]
    return f[order[first & last]].astype(np.int32)


def _tag_box_facets(pts, facets, lo, hi, tagger) -> np.ndarray:
    c = pts[facets]                      # (F,3,3)
    tol = 1e-9 * float(np.max(np.asarray(hi) - np.asarray(lo)))
    on = {}
    for ax, nm in enumerate("xyz"):
        on[nm + "lo"] = np.all(np.abs(c[:, :, ax] - lo[ax]) < tol, axis=1)
        on[nm + "hi"] = np.all(np.abs(c[:, :, ax] - hi[ax]) < tol, axis=1)
    return tagger(on, c.mean(axis=1)).astype(np.int32)


def duct_mesh(cells=(40, 10, 10), x_outlet: float = 4.0, *, jitter: float = 0.0,
              tags: dict | None = None, x_window=None) -> TetMesh:
    """Square duct [0,x_outlet] x [-.5,.5]^2 (DuctStokesFlow.py:36-124).

    With ``x_window=(c0, c1)`` only that range of cell planes is meshed (see box_tet_mesh); the cut
    planes are interior, so they carry no boundary facets."""
    tags = dict(DUCT_TAGS if tags is None else tags)
    lo, hi = (0.0, -0.5, -0.5), (float(x_outlet), 0.5, 0.5)
    pts, tets, _ = box_tet_mesh(lo, hi, cells, jitter=jitter, x_window=x_window)
    fac = _boundary_facets(tets)
    if x_window is not None:
        sx = (int(cells[1]) + 1) * (int(cells[2]) + 1)
        plane = fac // sx
        cut = np.zeros(len(fac), dtype=bool)
        if x_window[0] > 0:
            cut |= np.all(plane == 0, axis=1)
        if x_window[1] < int(cells[0]):
            cut |= np.all(plane == x_window[1] - x_window[0], axis=1)
        fac = fac[~cut]

    def tagger(on, _cent):
        t = np.full(fac.shape[0], tags["wall"])
        t[on["xlo"]] = tags["inlet"]
        t[on["xhi"]] = tags["outlet"]
        return t

    ft = _tag_box_facets(pts, fac, lo, hi, tagger)
    meta = {"cells": tuple(cells), "lo": lo, "hi": hi, "tags": tags, "kind": "duct"}
    if x_window is not None:
        meta["x_window"] = (int(x_window[0]), int(x_window[1]))
    return TetMesh(pts, tets, fac, ft, name="duct", meta=meta)


def channel_mesh(cells=(40, 10, 10), *, inner_half_width: float = 0.25, jitter: float = 0.0) -> TetMesh:
    """4x1x1 two-stream channel with the tag set of image2gmsh3D.py:435-438.

    The PNG->contour->OCC pipeline needs gmsh/skimage (absent offline); this is
    the synthetic stand-in of SURVEY 8 config 4: inlet facets whose centroid
    lies in the centred square of half width ``inner_half_width`` are the inner
    stream (inlet_1), the remaining inlet facets the outer stream (inlet_2).
    """
    tags = dict(CHANNEL_TAGS)
    lo, hi = (0.0, -0.5, -0.5), (4.0, 0.5, 0.5)
    pts, tets, _ = box_tet_mesh(lo, hi, cells, jitter=jitter)
    fac = _boundary_facets(tets)

    def tagger(on, cent):
        t = np.full(fac.shape[0], tags["wall"])
        inner = (np.abs(cent[:, 1]) < inner_half_width) & (np.abs(cent[:, 2]) < inner_half_width)
        t[on["xlo"] & inner] = tags["inlet_1"]
        t[on["xlo"] & ~inner] = tags["inlet_2"]
        t[on["xhi"]] = tags["outlet"]
        return t

    ft = _tag_box_facets(pts, fac, lo, hi, tagger)
    return TetMesh(pts, tets, fac, ft, name="channel",
                   meta={"cells": tuple(cells), "lo": lo, "hi": hi, "tags": tags, "kind": "channel",
                         "inner_half_width": inner_half_width})


def delaunay_duct_mesh(n: int = 8, x_outlet: float = 2.0, *, seed: int = 0, min_quality: float = 1e-9,
                       tags: dict | None = None) -> TetMesh:
    """Genuinely unstructured duct mesh: Delaunay tetrahedralisation (scipy) of a jittered point cloud with
    exact boundary points; flat tets (volume / h^3 < ``min_quality``) are dropped where that leaves the
    domain watertight (they sit on the boundary faces of the convex hull).  Variable valence, arbitrary cell
    orientation and vertex order -- the kind of input gmsh produces for the reference (DuctStokesFlow.py:36-124)."""
    from scipy.spatial import Delaunay
    tags = dict(DUCT_TAGS if tags is None else tags)
    rng = np.random.default_rng(seed)
    nx = max(2, int(round(n * x_outlet)))
    xs, ys, zs = np.linspace(0, x_outlet, nx + 1), np.linspace(-0.5, 0.5, n + 1), np.linspace(-0.5, 0.5, n + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    h = 1.0 / n
    I, J, K = np.meshgrid(np.arange(nx + 1), np.arange(n + 1), np.arange(n + 1), indexing="ij")
    # interior points move freely, points on a face only inside that face (edges / corners stay)
    d = rng.uniform(-0.3, 0.3, size=pts.shape) * h
    d[:, 0][((I == 0) | (I == nx)).ravel()] = 0.0
    d[:, 1][((J == 0) | (J == n)).ravel()] = 0.0
    d[:, 2][((K == 0) | (K == n)).ravel()] = 0.0
    pts = pts + d
    tets = Delaunay(pts).simplices.astype(np.int64)
    X4 = pts[tets]
    vol = np.abs(np.linalg.det(np.stack([X4[:, 1] - X4[:, 0], X4[:, 2] - X4[:, 0], X4[:, 3] - X4[:, 0]], axis=2))) / 6.0
    tets = tets[vol > min_quality * h ** 3]
    lo, hi = (0.0, -0.5, -0.5), (float(x_outlet), 0.5, 0.5)
    tets = tets.astype(np.int32)
    fac = _boundary_facets(tets)

    def tagger(on, _cent):
        t = np.full(fac.shape[0], tags["wall"])
        t[on["xlo"]] = tags["inlet"]
        t[on["xhi"]] = tags["outlet"]
        return t

    ft = _tag_box_facets(pts, fac, lo, hi, tagger)
    c = pts[fac]
    tol = 1e-9 * x_outlet
    on_hull = np.zeros(len(fac), dtype=bool)
    for ax in range(3):
        on_hull |= np.all(np.abs(c[:, :, ax] - lo[ax]) < tol, axis=1) | np.all(np.abs(c[:, :, ax] - hi[ax]) < tol, axis=1)
    if not on_hull.all():
        raise ValueError("delaunay_duct_mesh: dropping slivers opened the mesh; lower min_quality")
    if len(np.unique(tets)) != len(pts):
        raise ValueError("delaunay_duct_mesh: a point lost all its tets")
    return TetMesh(pts, tets, fac, ft, name="duct-delaunay",
                   meta={"lo": lo, "hi": hi, "tags": tags, "kind": "duct"})


def delaunay_channel_mesh(n: int = 16, *, length: float = 4.0, seed: int = 0, lattice: str = "bcc",
                          inner_half_width: float = 0.25) -> TetMesh:
    """Unstructured stand-in for the gmsh mesh of image2gmsh3D.py:445-486: Delaunay tetrahedralisation (scipy) of a
    slightly jittered body-centred (``lattice="bcc"``: nearly regular tets, what a production mesher delivers) or
    cubic (``"cubic"``: sliver-rich) lattice on the 4x1x1 channel, h = 1/n; facet tags of the two-stream channel
    (CHANNEL_TAGS; inlet facets by the position of their centroid, like ``channel_mesh``).  Nodes are ordered
    x-slowest (sorted by plane) so that x-slabs stay contiguous id ranges."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    h = 1.0 / n
    nx = int(round(length / h))
    xs, ys, zs = np.linspace(0, length, nx + 1), np.linspace(-0.5, 0.5, n + 1), np.linspace(-0.5, 0.5, n + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    I, J, K = np.meshgrid(np.arange(nx + 1), np.arange(n + 1), np.arange(n + 1), indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    d = rng.uniform(-0.25, 0.25, size=pts.shape) * h
    d[:, 0][((I == 0) | (I == nx)).ravel()] = 0.0
    d[:, 1][((J == 0) | (J == n)).ravel()] = 0.0
    d[:, 2][((K == 0) | (K == n)).ravel()] = 0.0
    if lattice == "bcc":
        d *= 0.2
        Xc, Yc, Zc = np.meshgrid(0.5 * (xs[1:] + xs[:-1]), 0.5 * (ys[1:] + ys[:-1]), 0.5 * (zs[1:] + zs[:-1]), indexing="ij")
        cen_pts = np.stack([Xc.ravel(), Yc.ravel(), Zc.ravel()], axis=1)
        cen_pts = cen_pts + rng.uniform(-0.05, 0.05, size=cen_pts.shape) * h
        pts = np.concatenate([pts + d, cen_pts])
    elif lattice == "cubic":
        pts = pts + d
    else:
        raise ValueError("lattice must be 'bcc' or 'cubic'")
    order = np.lexsort((pts[:, 2], pts[:, 1], np.round(pts[:, 0] / (0.5 * h))))       # x-slowest node ids
    pts = pts[order]
    tets = Delaunay(pts).simplices.astype(np.int64)
    X4 = pts[tets]
    vol = np.abs(np.linalg.det(np.stack([X4[:, 1] - X4[:, 0], X4[:, 2] - X4[:, 0], X4[:, 3] - X4[:, 0]], axis=2))) / 6.0
    tets = tets[vol > 1e-9 * h ** 3].astype(np.int32)
    if len(np.unique(tets)) != len(pts):
        raise ValueError("delaunay_channel_mesh: a point lost all its tets")
    fac = _boundary_facets(tets)
    lo, hi = (0.0, -0.5, -0.5), (float(length), 0.5, 0.5)
    tags = dict(CHANNEL_TAGS)

    def tagger(on, cent):
        t = np.full(fac.shape[0], -1)
        for k_ in ("ylo", "yhi", "zlo", "zhi"):
            t[on[k_]] = tags["wall"]
        inner = (np.abs(cent[:, 1]) < inner_half_width) & (np.abs(cent[:, 2]) < inner_half_width)
        t[on["xlo"] & inner] = tags["inlet_1"]
        t[on["xlo"] & ~inner] = tags["inlet_2"]
        t[on["xhi"]] = tags["outlet"]
        return t

    ft = _tag_box_facets(pts, fac, lo, hi, tagger)
    if (ft < 0).any():
        raise ValueError("delaunay_channel_mesh: dropping slivers opened the mesh")
    return TetMesh(pts, tets, fac, ft.astype(np.int32), name="channel-delaunay",
                   meta={"lo": lo, "hi": hi, "tags": tags, "kind": "channel", "inner_half_width": inner_half_width, "h": h})


DFG_TAGS = {"inlet": 2, "outlet": 3, "wall": 4, "obstacle": 5}   # DFG_3D_Validation.py:104-109


def dfg_pillar_mesh(n: int = 32, *, seed: int = 0, length: float = 2.2, width: float = 0.41, radius: float = 0.05,
                    centre=(0.5, 0.2), refine: float = 1.0, lattice: str = "bcc") -> TetMesh:
    """Channel [0,length] x [0,width]^2 with the circular pillar of dfg_pillar_3D.geo (r = 0.05 at (0.5, 0.2), axis
    along z): gmsh is not available offline, so the mesh is a Delaunay tetrahedralisation (scipy) of a slightly jittered
    body-centred (``lattice="bcc"``, default: nearly regular tets) or cubic lattice with h = width/n, exact points on the box faces and rings of points on the pillar surface
    (spacing h*refine); tets inside the pillar are removed.  Facet tags as in DFG_3D_Validation.py:104-109."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    h = width / n
    nx = int(round(length / h))
    xs, ys, zs = np.linspace(0, length, nx + 1), np.linspace(0, width, n + 1), np.linspace(0, width, n + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    I, J, K = np.meshgrid(np.arange(nx + 1), np.arange(n + 1), np.arange(n + 1), indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    d = rng.uniform(-0.25, 0.25, size=pts.shape) * h
    d[:, 0][((I == 0) | (I == nx)).ravel()] = 0.0
    d[:, 1][((J == 0) | (J == n)).ravel()] = 0.0
    d[:, 2][((K == 0) | (K == n)).ravel()] = 0.0
    if lattice == "bcc":
        # body-centred lattice: cube centres added, jitter reduced -- its Delaunay tets are nearly regular in the
        # interior (far fewer slivers than the perturbed cubic lattice)
        d *= 0.2
        Xc, Yc, Zc = np.meshgrid(0.5 * (xs[1:] + xs[:-1]), 0.5 * (ys[1:] + ys[:-1]), 0.5 * (zs[1:] + zs[:-1]), indexing="ij")
        cen_pts = np.stack([Xc.ravel(), Yc.ravel(), Zc.ravel()], axis=1)
        cen_pts = cen_pts + rng.uniform(-0.05, 0.05, size=cen_pts.shape) * h
        pts = np.concatenate([pts + d, cen_pts])
    else:
        pts = pts + d
    cx, cy = centre
    rr = np.hypot(pts[:, 0] - cx, pts[:, 1] - cy)
    pts = pts[rr > radius + 0.55 * h * refine]
    # rings on the pillar surface (and one ring just outside it) on every z-plane of the lattice, staggered
    nth = max(16, int(round(2 * np.pi * radius / (h * refine))))
    rings = []
    for k, z in enumerate(zs):
        for rad, off in ((radius, 0.0), (radius + 0.6 * h * refine, 0.5)):
            th = 2 * np.pi * (np.arange(nth) + 0.5 * (k % 2) + off) / nth
            rings.append(np.stack([cx + rad * np.cos(th), cy + rad * np.sin(th), np.full(nth, z)], axis=1))
    ring = np.concatenate(rings)
    n_lat = len(pts)
    pts = np.concatenate([pts, ring])
    on_pillar = np.zeros(len(pts), dtype=bool)
    on_pillar[n_lat:] = np.isclose(np.hypot(ring[:, 0] - cx, ring[:, 1] - cy), radius)
    tets = Delaunay(pts).simplices.astype(np.int64)
    X4 = pts[tets]
    vol = np.abs(np.linalg.det(np.stack([X4[:, 1] - X4[:, 0], X4[:, 2] - X4[:, 0], X4[:, 3] - X4[:, 0]], axis=2))) / 6.0
    cen = X4.mean(axis=1)
    inside = np.hypot(cen[:, 0] - cx, cen[:, 1] - cy) < radius * np.cos(np.pi / nth)
    inside |= on_pillar[tets].all(axis=1)                   # chord slivers spanned by surface points only
    tets = tets[(vol > 1e-9 * h ** 3) & ~inside].astype(np.int32)
    used = np.unique(tets)
    if len(used) != len(pts):                               # drop orphaned points, renumber
        remap = -np.ones(len(pts), dtype=np.int64)
        remap[used] = np.arange(len(used))
        pts, on_pillar, tets = pts[used], on_pillar[used], remap[tets].astype(np.int32)
    fac = _boundary_facets(tets)
    lo, hi = (0.0, 0.0, 0.0), (float(length), float(width), float(width))
    t = DFG_TAGS

    def tagger(on, _cent):
        tag = np.full(fac.shape[0], -1)
        for k_ in ("ylo", "yhi", "zlo", "zhi"):
            tag[on[k_]] = t["wall"]
        tag[on["xlo"]] = t["inlet"]
        tag[on["xhi"]] = t["outlet"]
        return tag

    ft = _tag_box_facets(pts, fac, lo, hi, tagger)
    pil = (ft < 0) & on_pillar[fac].all(axis=1)
    ft[pil] = t["obstacle"]
    if (ft < 0).any():
        raise ValueError(f"dfg_pillar_mesh: {(ft < 0).sum()} boundary facets are neither on the box nor on the pillar")
    return TetMesh(pts, tets, fac, ft.astype(np.int32), name="dfg-pillar",
                   meta={"lo": lo, "hi": hi, "tags": dict(t), "kind": "dfg", "radius": radius, "centre": tuple(centre),
                         "width": width, "length": length, "h": h})


def cavity_mesh(n: int = 16, *, jitter: float = 0.0) -> TetMesh:
    """Unit-cube lid-driven cavity, lid at y=1 (LidDrivenNavierStokesFlow.py:33-43 in 3-D)."""
    tags = dict(CAVITY_TAGS)
    lo, hi = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    pts, tets, _ = box_tet_mesh(lo, hi, (n, n, n), jitter=jitter)
    fac = _boundary_facets(tets)

    def tagger(on, _cent):
        t = np.full(fac.shape[0], tags["wall"])
        t[on["yhi"]] = tags["lid"]
        return t

    ft = _tag_box_facets(pts, fac, lo, hi, tagger)
    return TetMesh(pts, tets, fac, ft, name="cavity",
                   meta={"cells": (n, n, n), "lo": lo, "hi": hi, "tags": tags, "kind": "cavity"})


# --------------------------------------------------------------------------- #
# locality ordering for meshes that arrive in arbitrary order (gmsh)
# --------------------------------------------------------------------------- #
def _morton_key(points: np.ndarray, bits: int = 20) -> np.ndarray:
    lo, hi = points.min(axis=0), points.max(axis=0)
    q = ((points - lo) / np.maximum(hi - lo, 1e-300) * ((1 << bits) - 1)).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))


def reorder_for_locality(mesh: TetMesh):
    """Renumber nodes along a Morton (Z-order) curve and sort tets by their lowest node.

    gmsh writes nodes and cells in meshing order; the SpMV's x-gather and the assembly's
    gather lists want neighbours to be close in memory (data layout).  The
    CELL-LOCAL vertex order of every tet is untouched (the G metric depends on it).
    Returns (new_mesh, perm) with ``new_points = old_points[perm]``.
    """
    perm = np.argsort(_morton_key(mesh.points), kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    tets = inv[mesh.tets].astype(np.int32)
    order = np.argsort(tets.min(axis=1), kind="stable")
    facets = inv[mesh.facets].astype(np.int32) if len(mesh.facets) else mesh.facets
    new = TetMesh(np.ascontiguousarray(mesh.points[perm]), np.ascontiguousarray(tets[order]), facets,
                  mesh.facet_tags.copy(), name=mesh.name, meta=dict(mesh.meta))
    return new, perm


# --------------------------------------------------------------------------- #
# gmsh .msh reader (ASCII 2.2 and 4.1), tets (type 4) + triangles (type 2)
# --------------------------------------------------------------------------- #
def read_msh(path: str, reorder: bool = True) -> TetMesh:
    """Read tets and tagged boundary triangles from a gmsh ASCII file
    (nodes renumbered for locality unless ``reorder=False``).

    Mirrors what ``gmshio.model_to_mesh`` / ``read_from_msh`` deliver to the
    reference (NavierStokesChannelFlow.py:111, DFG_3D_Validation.py:79): only
    entities in physical groups; the triangle tag is the physical-group id.
    Cell-local vertex order is kept as written in the file.
    """
    with open(path, "r") as fh:
        lines = fh.read().split("\n")
    sec = {}
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        if ln.startswith("$") and not ln.startswith("$End"):
            nm = ln[1:]
            j = i + 1
            while lines[j].strip() != "$End" + nm:
                j += 1
            sec[nm] = lines[i + 1:j]
            i = j
        i += 1
    ver = float(sec["MeshFormat"][0].split()[0])
    if int(sec["MeshFormat"][0].split()[1]) != 0:
        raise ValueError("binary .msh files are not supported; write ASCII")
    if ver < 3.0:
        pts, ids, tris, tri_tags, tets = _read_msh2(sec)
    else:
        pts, ids, tris, tri_tags, tets = _read_msh4(sec)
    remap = -np.ones(int(ids.max()) + 1, dtype=np.int64)
    remap[ids] = np.arange(ids.size)
    tets = remap[tets].astype(np.int32)
    tris = remap[tris].astype(np.int32) if len(tris) else np.zeros((0, 3), np.int32)
    used = np.zeros(ids.size, dtype=bool)
    used[tets.ravel()] = True
    if not used.all():                                  # drop nodes no tet references
        new = -np.ones(ids.size, dtype=np.int64)
        new[used] = np.arange(int(used.sum()))
        pts, tets = pts[used], new[tets].astype(np.int32)
        keep = np.all(new[tris] >= 0, axis=1) if len(tris) else np.zeros(0, bool)
        tris, tri_tags = new[tris[keep]].astype(np.int32), np.asarray(tri_tags)[keep]
    out = TetMesh(np.ascontiguousarray(pts), np.ascontiguousarray(tets), np.ascontiguousarray(tris),
                  np.asarray(tri_tags, dtype=np.int32), name=path, meta={"kind": "msh", "version": ver})
    return reorder_for_locality(out)[0] if reorder else out


def _read_msh2(sec):
    nl = sec["Nodes"]
    n = int(nl[0])
    arr = np.array([ln.split() for ln in nl[1:1 + n]], dtype=np.float64)
    ids, pts = arr[:, 0].astype(np.int64), arr[:, 1:4]
    tris, tri_tags, tets = [], [], []
    el = sec["Elements"]
    for ln in el[1:1 + int(el[0])]:
        t = ln.split()
        et, ntag = int(t[1]), int(t[2])
        phys = int(t[3]) if ntag > 0 else 0
        nod = [int(v) for v in t[3 + ntag:]]
        if et == 2 and phys > 0:
            tris.append(nod)
            tri_tags.append(phys)
        elif et == 4:
            tets.append(nod)
    return pts, ids, np.array(tris, dtype=np.int64).reshape(-1, 3), tri_tags, np.array(tets, dtype=np.int64)


def _read_msh4(sec):
    ent_phys = {}                                       # (dim, tag) -> first physical tag
    if "Entities" in sec:
        el = sec["Entities"]
        npnt, ncur, nsur, nvol = (int(v) for v in el[0].split())
        k = 1 + npnt + ncur
        for dim, cnt in ((2, nsur), (3, nvol)):
            for ln in el[k:k + cnt]:
                t = ln.split()
                nphys = int(t[7])
                if nphys > 0:
                    ent_phys[(dim, int(t[0]))] = abs(int(t[8]))
            k += cnt
    nl = sec["Nodes"]
    nblocks, nnodes = int(nl[0].split()[0]), int(nl[0].split()[1])
    ids = np.empty(nnodes, dtype=np.int64)
    pts = np.empty((nnodes, 3), dtype=np.float64)
    k, w = 1, 0
    for _ in range(nblocks):
        cnt = int(nl[k].split()[3])
        ids[w:w + cnt] = [int(v) for v in nl[k + 1:k + 1 + cnt]]
        pts[w:w + cnt] = [[float(v) for v in ln.split()[:3]] for ln in nl[k + 1 + cnt:k + 1 + 2 * cnt]]
        k += 1 + 2 * cnt
        w += cnt
    tris, tri_tags, tets = [], [], []
    el = sec["Elements"]
    nblocks = int(el[0].split()[0])
    k = 1
    for _ in range(nblocks):
        dim, etag, et, cnt = (int(v) for v in el[k].split())
        rows = [[int(v) for v in ln.split()[1:]] for ln in el[k + 1:k + 1 + cnt]]
        if et == 2 and (2, etag) in ent_phys:
            tris += rows
            tri_tags += [ent_phys[(2, etag)]] * cnt
        elif et == 4:
            tets += rows
        k += 1 + cnt
    return pts, ids, np.array(tris, dtype=np.int64).reshape(-1, 3), tri_tags, np.array(tets, dtype=np.int64)


def write_msh2(mesh: TetMesh, path: str) -> None:
    """Write an ASCII gmsh-2.2 file (tests and mesh hand-over to other tools)."""
    with open(path, "w") as fh:
        fh.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % mesh.num_nodes)
        for i, p in enumerate(mesh.points):
            fh.write("%d %.17g %.17g %.17g\n" % (i + 1, p[0], p[1], p[2]))
        fh.write("$EndNodes\n$Elements\n%d\n" % (len(mesh.facets) + mesh.num_tets))
        e = 1
        for f, t in zip(mesh.facets, mesh.facet_tags):
            fh.write("%d 2 2 %d %d %d %d %d\n" % (e, t, t, f[0] + 1, f[1] + 1, f[2] + 1))
            e += 1
        for c in mesh.tets:
            fh.write("%d 4 2 1 1 %d %d %d %d\n" % (e, c[0] + 1, c[1] + 1, c[2] + 1, c[3] + 1))
            e += 1
        fh.write("$EndElements\n")
