"""Body-fitted tet mesh of the image-inlet channel (host side; numpy / scipy only).

What NavierStokes/image2gmsh3D.py:164-486 builds with gmsh's OpenCASCADE kernel: the box [0, 4] x [-.5, .5]^2 MINUS the
nozzle wall -- the band between the inner and the outer contour of the inlet image, extruded over x in [0, x_extrude = 0.5]
(:193-194) -- with
  * physical groups inlet_1 (inside the inner contour, x = 0) = 1, inlet_2 (between the outer contour and the duct wall,
    x = 0) = 2, outlet (x = 4) = 3, wall (duct walls, both nozzle surfaces, the band's end face at x = 0.5) = 4 (:435-438),
  * target sizes lc at the geometry points, and a background field = min of three Box fields (:445-483):
    0.75 lc for x in [-0.1, 0.25], 0.375 lc for x in [0.4, 0.6] (around the nozzle lip), 0.5 lc for x in [0.75, 1.0],
    2 lc elsewhere.
gmsh does not exist offline, and a Delaunay mesher without boundary recovery cannot honour the thin nozzle surfaces; but the
geometry is an EXTRUSION, which allows an exactly conforming construction from what is available:
  1. one triangulation of the cross-section [-.5, .5]^2 that contains BOTH contours as edge chains (the polygons are resampled,
     interior points kept off them: scipy's Delaunay triangulation of such a cloud contains the chains, and that is checked);
     its triangles are inner stream / band / outer stream by their centroid;
  2. node planes along x whose spacing follows the reference's size fields (graded between them), with planes exactly at
     x = 0, x_extrude and x_outlet;
  3. prisms over the fluid triangles for x < x_extrude (the band is left out: no node exists inside the wall) and over all
     triangles beyond, each cut into three tets along diagonals that are fixed by the vertex ids, so that neighbours agree.
The nozzle surfaces are therefore the contour polygons themselves, extruded -- the same surfaces the reference hands to
gmsh --, and the lip at x_extrude is a mesh plane.
  4. (second version) Behind the lip nothing needs the contours: from 0.15 behind it on the cross-section is the contour-free
     square lattice of the same spacing, and from where the planes lie more than 1.8 cells apart (the reference's far field,
     2 lc) the lattice of twice the spacing (1.5 lc in y, z: still finer than the reference's 2 lc), each change of
     cross-section through ONE layer of general tets (transition_slab: the 3-D Delaunay cells of the two planes' points,
     whose faces in either plane are that plane's triangles -- checked).  Away from the contours every prism is a Kuhn cell
     (right-triangle base, cut so that no dihedral angle exceeds 90 degrees) and the nodes are numbered along the cells' common
     diagonal: what the aggregation AMG needs (profiles/r5_prism_vs_kuhn.txt).
Difference to the reference's meshes, stated: in front of the far field the cross-section's size is 0.75 lc everywhere in
y, z, so the cells are flattened in x (0.5) around the lip where the reference refines isotropically to 0.375 lc.
"""
from __future__ import annotations

import numpy as np

from .inlet_contours import _resample_closed, points_in_polygon
from .mesh import TetMesh

CHANNEL_TAGS = {"inlet_1": 1, "inlet_2": 2, "outlet": 3, "wall": 4}       # image2gmsh3D.py:435-438


def size_along_x(x, lc: float, x_extrude: float = 0.5, growth: float = 0.35, far: float = 2.0):
    """Target size along the channel: the minimum of the reference's three Box fields (:445-483; VOut = 2 lc outside all of
    them), made continuous by letting the size grow by `growth` per unit length away from each box (gmsh grades the
    jump of a Box field without `Thickness` over a few cells in the same way)."""
    x = np.asarray(x, dtype=np.float64)
    boxes = ((-0.1, x_extrude - 0.25, 0.75 * lc), (x_extrude - 0.1, x_extrude + 0.1, 0.375 * lc),
             (x_extrude + 0.25, x_extrude + 0.5, 0.5 * lc))
    h = np.full_like(x, far * lc)
    for a, b, v in boxes:
        d = np.maximum(0.0, np.maximum(a - x, x - b))
        h = np.minimum(h, v + growth * d)
    return h


def x_planes(lc: float, x_extrude: float = 0.5, x_outlet: float = 4.0, far: float = 2.0) -> np.ndarray:
    """Node planes: marched with the local target size, each of the two stretches [0, x_extrude], [x_extrude, x_outlet]
    rescaled so that it ends exactly on its end plane."""
    def march(a, b):
        xs = [a]
        while xs[-1] < b:
            xs.append(xs[-1] + float(size_along_x(xs[-1], lc, x_extrude, far=far)))
        xs = np.array(xs)
        if len(xs) > 2 and (xs[-1] - b) > 0.5 * (xs[-1] - xs[-2]):
            xs = xs[:-1]                                       # the last step overshoots by more than half: drop it
        return a + (xs - a) * (b - a) / (xs[-1] - a)

    first = march(0.0, x_extrude)
    return np.concatenate([first, march(x_extrude, x_outlet)[1:]])


def resample_contour(poly: np.ndarray, h: float, corner_deg: float = 20.0) -> np.ndarray:
    """Points ON the closed polyline `poly` (m, 2), about h apart: the vertices where the polyline turns by more than
    `corner_deg` are kept (where two of them are closer than 0.6 h, the sharper one), the stretches between them are divided
    uniformly by arc length.  (Dividing every polygon edge for itself, as the 2-D inlet meshes do, leaves point pairs much
    closer than h wherever the RDP polygon has a short edge -- harmless for a Poisson solve, needle cells in an extrusion.)"""
    P = np.asarray(poly, dtype=np.float64)
    m = len(P)
    d_prev, d_next = P - np.roll(P, 1, axis=0), np.roll(P, -1, axis=0) - P
    ang = np.degrees(np.abs(np.arctan2(d_prev[:, 0] * d_next[:, 1] - d_prev[:, 1] * d_next[:, 0], (d_prev * d_next).sum(axis=1))))
    seg = np.linalg.norm(d_next, axis=1)                        # length of edge k -> k + 1
    s = np.concatenate([[0.0], np.cumsum(seg)])                 # arc length at vertex k (s[m] = perimeter)
    per = s[-1]
    corners = [k for k in np.argsort(-ang) if ang[k] > corner_deg]
    keep = []
    for k in corners:                                           # sharpest first; drop corners too close to a kept one
        if all(min(abs(s[k] - s[j]), per - abs(s[k] - s[j])) > 0.6 * h for j in keep):
            keep.append(k)
    if not keep:
        keep = [0]
    brk = np.sort(s[np.array(keep)])

    def at(t):                                                  # point at arc length t (mod perimeter)
        t = np.mod(t, per)
        k = np.minimum(np.searchsorted(s, t, side="right") - 1, m - 1)
        f = (t - s[k]) / np.where(seg[k] > 0, seg[k], 1.0)
        return P[k] + f[:, None] * (P[(k + 1) % m] - P[k])

    out = []
    for a, b in zip(brk, np.concatenate([brk[1:], [brk[0] + per]])):
        n = max(1, int(np.ceil((b - a) / h - 1e-9)))           # spacing <= h: see the exclusion distance in cross_section
        out.append(at(a + (b - a) * np.arange(n) / n))
    return np.concatenate(out)


def cross_section(contour_inner, contour_outer, h2: float, lattice: str = "square"):
    """Triangulation of [-.5, .5]^2 conforming to both contours.  Polygons are (m, 2) in (y, z).  Returns
    (points (n, 2), tris (e, 3), region (e,): 1 inner stream, 0 band, 2 outer stream, the two contour chains as meshed).

    Interior points: a SQUARE lattice aligned with the duct (``lattice="square"``, default), every cell cut along the same
    diagonal, and the points numbered by (z ascending, y descending).  The prisms over such right triangles, cut along
    diagonals that follow the numbering, are Kuhn cells -- no dihedral angle above 90 degrees, where an equilateral base gives
    104-117 -- and the numbering runs along the cells' common diagonal, which is what the greedy aggregation of the AMG
    hierarchy needs to find cube-shaped aggregates (profiles/r5_prism_vs_kuhn.txt: both together are worth a third of the
    Krylov iterations on a plain duct).  Only the strips along the contours hold general triangles.  ``lattice="hex"``: the
    hexagonal lattice of the first version."""
    from scipy.spatial import Delaunay, cKDTree
    square = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])
    nx = max(2, int(round(1.0 / h2)))
    # (square lattice: the duct wall's points ARE lattice points, so the right triangles reach the wall)
    chains = [resample_contour(np.asarray(contour_inner, dtype=np.float64), h2),
              resample_contour(np.asarray(contour_outer, dtype=np.float64), h2),
              _resample_closed(square, 1.0 / nx if lattice == "square" else h2)]
    bpts = np.concatenate(chains)
    if lattice == "square":
        jj, ii = np.meshgrid(np.arange(nx + 1), np.arange(nx + 1), indexing="ij")
        lat = np.stack([-0.5 + ii / nx, -0.5 + jj / nx], axis=-1).reshape(-1, 2)
    else:
        ny = max(2, int(round(1.0 / (h2 * np.sqrt(3.0) / 2.0))))
        jj, ii = np.meshgrid(np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
        lat = np.stack([-0.5 + (ii + 0.5 * (jj % 2)) / nx, -0.5 + jj / ny], axis=-1).reshape(-1, 2)
    lat = lat[(np.abs(lat) < 0.5 - 1e-9).all(axis=1)]
    # interior points stay 0.6 h2 away from the chains THEMSELVES (measured to a 4x finer sampling of them): every chain segment is
    # at most h2 long, so its diametral circle (radius <= h2 / 2) is empty and the segment is an edge of the Delaunay triangulation
    fine = np.concatenate([resample_contour(c, 0.25 * h2, corner_deg=0.0) if len(c) > 4 else _resample_closed(c, 0.25 * h2)
                           for c in (np.asarray(contour_inner, dtype=np.float64), np.asarray(contour_outer, dtype=np.float64), square)])
    d_c, _ = cKDTree(fine[:-len(_resample_closed(square, 0.25 * h2))] if lattice == "square" else fine).query(lat)
    lat = lat[d_c > 0.6 * h2]                                  # (square lattice: only the two contours exclude, the wall is part of it)
    pts = np.concatenate([bpts, lat])
    # a tiny shear decides the ties of the square cells (four cocircular points) the same way everywhere: the shorter diagonal of
    # every sheared cell is the one from (y + 1, z) to (y, z + 1)
    shear = pts + np.stack([1e-4 * pts[:, 1], np.zeros(len(pts))], axis=1) if lattice == "square" else pts
    tris = Delaunay(shear).simplices.astype(np.int64)
    a = pts[tris]
    area2 = np.abs((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    tris = tris[area2 > 1e-3 * h2 * h2]                      # (qhull leaves needle triangles between collinear points of the square's edges)
    # conformity: every segment of the two contour chains must be an edge of the triangulation
    edges = set()
    for t in tris:
        for u, v in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            edges.add((min(u, v), max(u, v)))
    off = 0
    for c in chains[:2]:
        n = len(c)
        for k in range(n):
            u, v = off + k, off + (k + 1) % n
            if (min(u, v), max(u, v)) not in edges:
                raise ValueError("cross-section triangulation does not contain the nozzle contour: choose a smaller mesh size "
                                 "(the contours come closer to each other or to the duct wall than the cross-section size)")
        off += n
    cen = pts[tris].mean(axis=1)
    inside_in = points_in_polygon(cen, chains[0])                # (the chains ARE the mesh's nozzle surfaces)
    inside_out = points_in_polygon(cen, chains[1])
    region = np.where(inside_in, 1, np.where(inside_out, 0, 2)).astype(np.int8)
    if lattice == "square":
        # number the points by (z ascending, y descending): the right angle of every lattice triangle then sits at its MIDDLE
        # vertex, which is what makes the three tets of its prism Kuhn cells (nozzle_channel_mesh cuts along the numbering)
        order = np.lexsort((-np.round(pts[:, 0], 12), np.round(pts[:, 1], 12)))
        new = np.empty(len(pts), dtype=np.int64)
        new[order] = np.arange(len(pts))
        pts, tris = pts[order], new[tris]
    return pts, tris, region, chains[0], chains[1]


def lattice_section(nx: int):
    """The contour-free cross-section: the full square lattice of [-.5, .5]^2 with nx cells a side, every cell cut along the
    diagonal from (y + 1, z) to (y, z + 1), points numbered by (z ascending, y descending) -- the interior of ``cross_section``'s
    square lattice continued up to the contours' place.  Returns (points (n, 2), tris (e, 3))."""
    jj, ii = np.meshgrid(np.arange(nx + 1), np.arange(nx + 1), indexing="ij")
    pts = np.stack([-0.5 + ii / nx, -0.5 + jj / nx], axis=-1).reshape(-1, 2)
    order = np.lexsort((-np.round(pts[:, 0], 12), np.round(pts[:, 1], 12)))
    new = np.empty(len(pts), dtype=np.int64)
    new[order] = np.arange(len(pts))
    node = lambda i, j: new[j * (nx + 1) + i]
    i, j = np.meshgrid(np.arange(nx), np.arange(nx), indexing="ij")
    i, j = i.ravel(), j.ravel()
    tris = np.concatenate([np.stack([node(i + 1, j), node(i, j), node(i, j + 1)], axis=1),
                           np.stack([node(i + 1, j), node(i + 1, j + 1), node(i, j + 1)], axis=1)])
    return pts[order], tris.astype(np.int64)


def transition_slab(p_bot, t_bot, x_bot: float, p_top, t_top, x_top: float, h2: float):
    """Tets of the one layer between two DIFFERENT cross-sections (both Delaunay triangulations of their points under the shear
    of ``cross_section``): the 3-D Delaunay tetrahedralisation of the two planes' points.  Its faces in either plane are the
    2-D Delaunay triangles of that plane's points, i.e. the layer conforms to the prisms below and above -- checked, not assumed.
    qhull sees the points through a linear map close to the identity (the in-plane shear that decides the lattice cells'
    diagonals, and the upper plane shifted by a few per cent of a cell, so that no eight lattice points are cospherical): the
    result is a valid triangulation of the true points, only not exactly their Delaunay one.  Returns (tets (e, 4) with bottom
    nodes 0..nb-1 and top nodes nb.., smallest cell volume / mean cell volume)."""
    from scipy.spatial import Delaunay
    nb, nt = len(p_bot), len(p_top)
    P = np.zeros((nb + nt, 3))
    P[:nb, 0], P[:nb, 1:] = x_bot, p_bot
    P[nb:, 0], P[nb:, 1:] = x_top, p_top
    Q = P.copy()
    Q[:, 1] += 1e-4 * P[:, 2]
    Q[nb:, 1] += 0.031 * h2
    Q[nb:, 2] += 0.017 * h2
    T = Delaunay(Q).simplices.astype(np.int64)
    nbot = (T < nb).sum(axis=1)
    if ((nbot == 0) | (nbot == 4)).any():
        raise ValueError("transition layer: a cell lies in one plane")
    vol = np.einsum("ij,ij->i", np.cross(P[T[:, 1]] - P[T[:, 0]], P[T[:, 2]] - P[T[:, 0]]), P[T[:, 3]] - P[T[:, 0]]) / 6.0
    T[vol < 0] = T[vol < 0][:, [0, 2, 1, 3]]
    vol = np.abs(vol)
    if abs(vol.sum() - (x_top - x_bot)) > 1e-9:
        raise ValueError("transition layer: the cells do not fill the slab")

    def plane_faces(sel_count, lower):
        t = np.sort(T[nbot == sel_count], axis=1)              # three nodes of one plane + one of the other
        f = t[:, :3] if lower else t[:, 1:] - nb
        return np.unique(f, axis=0)

    if not (np.array_equal(plane_faces(3, True), np.unique(np.sort(t_bot, axis=1), axis=0)) and
            np.array_equal(plane_faces(1, False), np.unique(np.sort(t_top, axis=1), axis=0))):
        raise ValueError("transition layer: its faces are not the cross-sections' triangles")
    return T, float(vol.min() / vol.mean())


def nozzle_channel_mesh(contour_inner, contour_outer, lc: float, *, x_extrude: float = 0.5, x_outlet: float = 4.0,
                        cross_size: float | None = None, far: float = 2.0, lattice_behind: float | None = 0.15,
                        coarse_far_field: bool = True) -> TetMesh:
    """The body-fitted channel (module docstring).  contour_* are (m, 2) polygons in (y, z); lc is the reference's
    ``channel_mesh_size`` (NavierStokesChannelFlow.py:81-93); the cross-section is triangulated at ``cross_size``
    (default 0.75 lc: what the reference's inlet region gets -- min of the points' lc and the first Box field's 0.75 lc,
    :445-455 -- and, measured, the shape the aggregation AMG likes best: cells of aspect 2.7 in the far field instead
    of 4 with lc / 2)."""
    h2 = float(cross_size) if cross_size else 0.75 * float(lc)
    p2, tris, region, chain_in, chain_out = cross_section(contour_inner, contour_outer, h2)
    xs = x_planes(float(lc), x_extrude, x_outlet, far)
    k_lip = int(np.argmin(np.abs(xs - x_extrude)))
    n2, nl = len(p2), len(xs)
    fluid_tri = region != 0
    fluid_node = np.zeros(n2, dtype=bool)
    fluid_node[tris[fluid_tri].ravel()] = True
    # Behind the lip nothing needs the contours, and the strips of general triangles along them cost Krylov iterations as long as
    # they run (profiles/r5_prism_vs_kuhn.txt, case (h): 1.5x on a plain inflow): from the first plane at least `lattice_behind`
    # behind the lip on, the cross-section is the contour-free lattice of the same spacing; and from where the planes lie more
    # than 1.8 cross-section cells apart (the reference's far field, 2 lc, :445-483) the lattice of TWICE the spacing (1.5 lc: still
    # finer than the reference's far field), so that the far-field cells are not stretched.  ONE layer of general tets joins two
    # different cross-sections (transition_slab); a transition that cannot be made well leaves the cross-section as it is.
    nx = max(2, int(round(1.0 / h2)))
    sections = [(p2, tris)]                                      # (points, triangles) of the cross-sections, in the order they appear
    sec_of = np.zeros(nl, dtype=np.int64)                        # plane -> cross-section
    slabs = {}                                                   # lower plane of a transition layer -> its tets
    quality = []
    if lattice_behind is not None:
        wanted = [(lattice_section(nx), np.nonzero(xs >= x_extrude + float(lattice_behind) - 1e-12)[0])]
        if coarse_far_field and nx >= 8:
            dx = np.diff(xs)
            wanted.append((lattice_section((nx + 1) // 2), np.nonzero(dx >= 1.8 / nx)[0]))
        k_prev = k_lip
        for (pS, tS), cand in wanted:
            cand = cand[(cand > k_prev) & (cand < nl - 2)]
            if not len(cand):
                break
            k = int(cand[0])
            pL, tL = sections[-1]
            try:
                slab, q = transition_slab(pL, tL, xs[k], pS, tS, xs[k + 1], h2)
            except ValueError:
                break
            if q < 1e-3:                                         # (a sliver thinner than 1/1000 of the mean cell: keep the section)
                break
            sections.append((pS, tS))
            sec_of[k + 1:] = len(sections) - 1
            slabs[k] = slab
            quality.append(q)
            k_prev = k + 1
    # node numbering: plane by plane; planes upstream of the lip hold the fluid nodes only (nothing exists inside the wall)
    width = max(len(ps) for ps, _ in sections)
    ids = -np.ones((nl, width), dtype=np.int64)
    count = 0
    for k in range(nl):
        n_k = len(sections[sec_of[k]][0])
        sel = np.zeros(width, dtype=bool)
        sel[:n_k] = fluid_node if (k < k_lip and sec_of[k] == 0) else True
        ids[k, sel] = count + np.arange(int(sel.sum()))
        count += int(sel.sum())
    pts = np.zeros((count, 3))
    for k in range(nl):
        ps = sections[sec_of[k]][0]
        sel = ids[k, :len(ps)] >= 0
        pts[ids[k, :len(ps)][sel], 0] = xs[k]
        pts[ids[k, :len(ps)][sel], 1:] = ps[sel]
    sorted_tris = [np.sort(ts_, axis=1) for _, ts_ in sections]  # v0 < v1 < v2: the diagonals of the quads follow the ids
    tets = []
    for k in range(nl - 1):
        if k in slabs:                                           # the one layer of general tets between two cross-sections
            nb_, nt_ = len(sections[sec_of[k]][0]), len(sections[sec_of[k + 1]][0])
            both = np.concatenate([ids[k, :nb_], ids[k + 1, :nt_]])
            tets.append(both[slabs[k]])
            continue
        t = sorted_tris[sec_of[k]]
        if k < k_lip and sec_of[k] == 0:
            t = t[fluid_tri]
        b0, b1, b2 = ids[k, t[:, 0]], ids[k, t[:, 1]], ids[k, t[:, 2]]
        w0, w1, w2 = ids[k + 1, t[:, 0]], ids[k + 1, t[:, 1]], ids[k + 1, t[:, 2]]
        tets.append(np.concatenate([np.stack([b0, b1, b2, w2], axis=1), np.stack([b0, b1, w2, w1], axis=1),
                                    np.stack([b0, w0, w1, w2], axis=1)]))
    tets = np.concatenate(tets)
    if tets.min() < 0:
        raise RuntimeError("internal: a prism refers to a node inside the nozzle wall")
    # boundary facets = faces of exactly one tet, tagged by where they lie
    from .mesh import _boundary_facets
    facets = _boundary_facets(tets.astype(np.int32))
    cen = pts[facets].mean(axis=1)
    on_x = np.ptp(pts[facets][:, :, 0], axis=1) < 1e-12
    T = CHANNEL_TAGS
    ftags = np.full(len(facets), T["wall"], dtype=np.int32)
    at_in = on_x & (np.abs(cen[:, 0]) < 1e-9)
    yz = cen[:, 1:]
    inner = points_in_polygon(yz, chain_in)
    ftags[at_in & inner] = T["inlet_1"]
    ftags[at_in & ~inner] = T["inlet_2"]                          # (no facet of the band exists at x = 0: it is not meshed)
    ftags[on_x & (np.abs(cen[:, 0] - x_outlet) < 1e-9)] = T["outlet"]
    meta = {"kind": "channel-nozzle", "tags": dict(T), "lc": float(lc), "cross_size": h2, "x_extrude": float(x_extrude),
            "planes": int(nl), "lip_plane": k_lip, "cross_section_nodes": int(n2), "cross_section_triangles": int(len(tris)),
            "band_triangles": int((~fluid_tri).sum()),
            "transition_planes": sorted(int(k) for k in slabs), "transition_x": [float(xs[k]) for k in sorted(slabs)],
            "transition_smallest_cell": [float(q) for q in quality],
            "cross_sections": [{"nodes": int(len(ps)), "triangles": int(len(ts_))} for ps, ts_ in sections]}
    return TetMesh(pts, tets.astype(np.int32), facets.astype(np.int32), ftags, name="nozzle-channel", meta=meta)


def channel_from_image_bodyfitted(img_fname: str, flowrate_ratio: float, lc: float, *, max_pixels: int | None = 1024,
                                  cross_size: float | None = None, far: float = 2.0, x_extrude: float = 0.5):
    """(mesh, (mask, g), inlet profiles) of NavierStokesChannelFlow.py's fine or coarse stage on the body-fitted channel:
    generate_inlet_profiles + generate_mesh + create_boundary_conditions (:102-147).  Dirichlet sets in the reference's order
    [wall, inlet_1, inlet_2, outlet] (:146: later entries win on shared nodes): wall u = 0, inlets u = (profile, 0, 0) by
    the non-matching interpolation of the 2-D Poisson profiles (:150-157), outlet p = 0."""
    from .bcs import DirichletBC, DirichletSet
    from .inlet_contours import solve_inlet_profiles
    data = solve_inlet_profiles(img_fname, flowrate_ratio, max_pixels=max_pixels)
    m = nozzle_channel_mesh(data.contour_inner[:, ::-1], data.contour_outer[:, ::-1], lc, cross_size=cross_size, far=far,
                            x_extrude=x_extrude)
    t = m.meta["tags"]
    pts = m.points

    def bc(nodes, comps, vals):
        return DirichletBC(np.asarray(nodes, dtype=np.int64), comps, np.asarray(vals, dtype=np.float64))

    def vec(profile, x):
        return np.stack([profile(x), np.zeros(len(x)), np.zeros(len(x))], axis=1)

    wall, n1, n2, out = (m.facet_nodes(t[k]) for k in ("wall", "inlet_1", "inlet_2", "outlet"))
    bcs = DirichletSet(m, [bc(wall, (0, 1, 2), np.zeros((len(wall), 3))),
                           bc(n1, (0, 1, 2), vec(data.profile_1, pts[n1])),
                           bc(n2, (0, 1, 2), vec(data.profile_2, pts[n2])),
                           bc(out, (3,), np.zeros((len(out), 1)))])
    return m, bcs.flatten(), data


def inlet_fluxes(mesh: TetMesh, w) -> tuple[float, float, float]:
    """(flux through inlet_1, inlet_2, outlet) of the P1 velocity in the dof vector w: int u.n over the tagged facets
    (n = +x on all three: into the channel at the inlets, out of it at the outlet)."""
    W = np.asarray(w, dtype=np.float64).reshape(-1, 4)
    out = []
    for key in ("inlet_1", "inlet_2", "outlet"):
        f = mesh.facets[mesh.facet_tags == mesh.meta["tags"][key]]
        X = mesh.points[f]
        area = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
        out.append(float(np.sum(area * W[f, 0].mean(axis=1))))
    return tuple(out)
