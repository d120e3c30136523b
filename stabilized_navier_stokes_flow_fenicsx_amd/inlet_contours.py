"""Inlet image -> contours -> two 2-D P1 Poisson profiles, the way NavierStokes/image2inlet.py does it
(host side; numpy / scipy / PIL only -- skimage, rdp, shapely and gmsh do not exist offline):

  load_image          image2inlet.py:42-56   RGBA over white, skimage's rgb2gray luma weights
  find_contours       :58-61                 marching squares at level 0.5 with linear interpolation along the grid
                                             edges (what ``skimage.measure.find_contours`` computes), closed polylines
                                             in (row, col) coordinates, first point repeated at the end
  get_contours        :63-91                 area filter: the contour's rounded pixels, holes filled, must cover
                                             >= 5 % of the image (:66-76); then the normalisation of :86-91
  optimize_contour    :94-139                low-pass of x + iy in Fourier space (|freq| > 0.12 zeroed, :105-109),
                                             Ramer-Douglas-Peucker with epsilon = 5e-4 (:118), closing point dropped
                                             (:122), mesh_lc = 0.05 * smaller extent (:135-137)
  region meshes       :141-232               inner: the polygon; outer: the unit square [-.5,.5]^2 minus the polygon.
                                             gmsh is replaced by boundary points at spacing mesh_lc + a hexagonal
                                             lattice inside, triangulated by scipy's Delaunay (cells outside dropped)
  solve_velocity_field :240-291              P1 Poisson  (grad u, grad v) = (10, v),  u = 0 on every boundary edge,
                                             sparse LU; area = int 1, average = int u / area by exact P1 quadrature
  solve_inlet_profiles :294-353              u /= average; u *= ratio / area  resp. (1 - ratio) / area

Coordinates as the reference hands them to gmsh (:165, :206): y = contour[:, 1], z = contour[:, 0].
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.ndimage as ndi
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def load_image(path: str) -> np.ndarray:
    """Grayscale in [0, 1] at the image's own resolution."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("RGBA", "LA", "P"):
        im = im.convert("RGBA")
        a = np.asarray(im, dtype=np.float64) / 255.0
        rgb = a[..., :3] * a[..., 3:4] + (1.0 - a[..., 3:4])          # rgba2rgb: white background
    elif im.mode == "L":
        return np.asarray(im, dtype=np.float64) / 255.0
    else:
        rgb = np.asarray(im.convert("RGB"), dtype=np.float64) / 255.0
    return 0.2125 * rgb[..., 0] + 0.7154 * rgb[..., 1] + 0.0721 * rgb[..., 2]


# --------------------------------------------------------------------------------------------------------------
# marching squares
# --------------------------------------------------------------------------------------------------------------
def find_contours(img: np.ndarray, level: float = 0.5) -> list:
    """Iso-contours of ``img`` at ``level``: list of (n, 2) arrays of (row, col) points, one point per crossed grid
    edge, linear interpolation; closed contours repeat their first point.  Ambiguous (saddle) cells connect the
    LOW corners ('fully_connected = low', skimage's default)."""
    a = np.asarray(img, dtype=np.float64)
    hi = a > level
    H, W = a.shape
    # crossings on horizontal edges (r, c)-(r, c+1) and vertical edges (r, c)-(r+1, c); edge ids: h: r*W + c,
    # v: H*W + r*W + c
    def lerp(v0, v1):
        return (level - v0) / (v1 - v0)

    tl, tr, bl, br = hi[:-1, :-1], hi[:-1, 1:], hi[1:, :-1], hi[1:, 1:]
    case = tl.astype(np.int8) | (tr.astype(np.int8) << 1) | (bl.astype(np.int8) << 2) | (br.astype(np.int8) << 3)
    rr, cc = np.nonzero((case != 0) & (case != 15))
    cs = case[rr, cc]
    top = rr * W + cc
    bot = (rr + 1) * W + cc
    left = H * W + rr * W + cc
    right = H * W + rr * W + cc + 1
    segs = []
    # per case: pairs of cell edges joined by a segment (T, B, L, R)
    table = {1: [("T", "L")], 2: [("T", "R")], 3: [("L", "R")], 4: [("L", "B")], 5: [("T", "B")],
             6: [("T", "R"), ("L", "B")],            # saddle: tr, bl high, low corners (tl, br) stay connected
             7: [("B", "R")], 8: [("B", "R")],
             9: [("T", "L"), ("B", "R")],            # saddle: tl, br high, low corners (tr, bl) stay connected
             10: [("T", "B")], 11: [("L", "B")], 12: [("L", "R")], 13: [("T", "R")], 14: [("T", "L")]}
    edge = {"T": top, "B": bot, "L": left, "R": right}
    for cval, pairs in table.items():
        sel = cs == cval
        if not sel.any():
            continue
        for e0, e1 in pairs:
            segs.append(np.stack([edge[e0][sel], edge[e1][sel]], axis=1))
    if not segs:
        return []
    segs = np.concatenate(segs)
    # coordinates of an edge id
    def coords(eid):
        eid = np.asarray(eid)
        horiz = eid < H * W
        e = np.where(horiz, eid, eid - H * W)
        r, c = e // W, e % W
        out = np.empty((len(eid), 2))
        hr, hc = r[horiz], c[horiz]
        t = lerp(a[hr, hc], a[hr, hc + 1])
        out[horiz, 0] = hr
        out[horiz, 1] = hc + t
        vr, vc = r[~horiz], c[~horiz]
        t = lerp(a[vr, vc], a[vr + 1, vc])
        out[~horiz, 0] = vr + t
        out[~horiz, 1] = vc
        return out

    # link segments: every edge id is shared by at most two segments
    nbr = {}
    for e0, e1 in segs.tolist():
        nbr.setdefault(e0, []).append(e1)
        nbr.setdefault(e1, []).append(e0)
    seen = set()
    contours = []
    starts = sorted(nbr)                                  # raster order: the outermost contour is met first
    # open contours (ending on the image border) first need their end points as starts
    ends = [e for e in starts if len(nbr[e]) == 1]
    for s in ends + starts:
        if s in seen:
            continue
        path = [s]
        seen.add(s)
        cur, prev = s, None
        while True:
            nxt = [n for n in nbr[cur] if n != prev or nbr[cur].count(n) > 1]
            nxt = [n for n in nxt if n not in seen]
            if not nxt:
                break
            prev, cur = cur, nxt[0]
            path.append(cur)
            seen.add(cur)
        closed = len(nbr[s]) == 2 and s in nbr[cur] and len(path) > 2
        pts = coords(np.array(path))
        if closed:
            pts = np.concatenate([pts, pts[:1]])
        contours.append(pts)
    return contours


def get_contours(gray: np.ndarray) -> list:
    """get_contours (:58-91): contours covering >= 5 % of the image, normalised to the unit square."""
    height, width = gray.shape
    out = []
    for contour in find_contours(gray, 0.5):
        mask = np.zeros(gray.shape, dtype=bool)
        mask[np.round(contour[:, 0]).astype(int), np.round(contour[:, 1]).astype(int)] = True
        mask = ndi.binary_fill_holes(mask)
        if float(np.count_nonzero(mask)) / float(height * width) >= 0.05:
            c = contour.copy()
            c[:, 1] -= 0.5 * height                       # (sic) the reference scales columns by the height ...
            c[:, 1] /= height
            c[:, 0] -= 0.5 * width                        # ... and rows by the width (:86-91); square images
            c[:, 0] /= width
            c[:, 0] *= -1.0
            out.append(c)
    return out


def rdp(points: np.ndarray, epsilon: float) -> np.ndarray:
    """Ramer-Douglas-Peucker (the ``rdp`` package's semantics: end points kept; for coinciding end points the
    distance to that point)."""
    n = len(points)
    keep = np.zeros(n, dtype=bool)
    keep[0] = keep[-1] = True
    stack = [(0, n - 1)]
    while stack:
        i0, i1 = stack.pop()
        if i1 <= i0 + 1:
            continue
        p0, p1 = points[i0], points[i1]
        seg = points[i0 + 1:i1]
        d = p1 - p0
        L = float(np.hypot(d[0], d[1]))
        if L == 0.0:
            dist = np.hypot(seg[:, 0] - p0[0], seg[:, 1] - p0[1])
        else:
            dist = np.abs(d[0] * (seg[:, 1] - p0[1]) - d[1] * (seg[:, 0] - p0[0])) / L
        k = int(np.argmax(dist))
        if dist[k] > epsilon:
            keep[i0 + 1 + k] = True
            stack.append((i0, i0 + 1 + k))
            stack.append((i0 + 1 + k, i1))
    return points[keep]


def optimize_contour(contour: np.ndarray, cutoff: float = 0.12, epsilon: float = 0.0005):
    """optimize_contour (:94-139) -> (polygon vertices (m, 2) in (z, y) = contour columns (0, 1) order, mesh_lc)."""
    c = np.array(contour, dtype=np.float64)
    signal = c[:, 1] + 1j * c[:, 0]
    f = np.fft.fft(signal)
    freq = np.fft.fftfreq(signal.shape[-1])
    f[np.abs(freq) > cutoff] = 0
    filt = np.fft.ifft(f)
    c[:, 1] = filt.real
    c[:, 0] = filt.imag
    c = rdp(c, epsilon)
    c = np.delete(c, len(c) - 1, 0)                        # the closing point coincides with the first (:122)
    ext = min(c[:, 1].max() - c[:, 1].min(), c[:, 0].max() - c[:, 0].min())
    return c, 0.05 * ext


# --------------------------------------------------------------------------------------------------------------
# polygon regions -> triangles -> P1 Poisson
# --------------------------------------------------------------------------------------------------------------
def points_in_polygon(p: np.ndarray, poly: np.ndarray) -> np.ndarray:
    """Even-odd rule, vectorised over the points; poly (m, 2) open polygon."""
    x, y = p[:, 0], p[:, 1]
    inside = np.zeros(len(p), dtype=bool)
    x0, y0 = poly[-1]
    for x1, y1 in poly:
        cond = (y0 > y) != (y1 > y)
        with np.errstate(divide="ignore", invalid="ignore"):
            xi = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
        inside ^= cond & (x < xi)
        x0, y0 = x1, y1
    return inside


def _resample_closed(poly: np.ndarray, lc: float) -> np.ndarray:
    out = []
    for a, b in zip(poly, np.roll(poly, -1, axis=0)):
        n = max(1, int(np.ceil(np.hypot(*(b - a)) / lc)))
        t = np.arange(n)[:, None] / n
        out.append(a + t * (b - a))
    return np.concatenate(out)


@dataclass
class RegionSolution:
    points: np.ndarray          # (n, 2) as (y, z)
    tris: np.ndarray            # (e, 3)
    u: np.ndarray               # nodal values (scaled)
    area: float
    average_raw: float          # int u / area of the unscaled Poisson solution

    def __post_init__(self):
        from scipy.spatial import Delaunay
        self._tri = Delaunay(self.points)
        key = {tuple(sorted(t)) for t in self.tris.tolist()}
        self._kept = np.array([tuple(sorted(t)) in key for t in self._tri.simplices.tolist()])

    def integral(self) -> float:
        a = self.points[self.tris]
        ar = 0.5 * np.abs((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) -
                          (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
        return float(np.sum(ar * self.u[self.tris].mean(axis=1)))

    def contains(self, y, z) -> np.ndarray:
        s = self._tri.find_simplex(np.stack([np.asarray(y, float), np.asarray(z, float)], axis=1))
        return (s >= 0) & self._kept[np.maximum(s, 0)]

    def __call__(self, y, z) -> np.ndarray:
        """P1 interpolant at (y, z); 0 outside the region (the reference's non-matching interpolation, :150-157)."""
        q = np.stack([np.asarray(y, float), np.asarray(z, float)], axis=1)
        s = self._tri.find_simplex(q)
        ok = (s >= 0) & self._kept[np.maximum(s, 0)]
        out = np.zeros(len(q))
        if ok.any():
            T = self._tri.transform[s[ok]]
            b = np.einsum("nij,nj->ni", T[:, :2], q[ok] - T[:, 2])
            bary = np.concatenate([b, 1 - b.sum(axis=1, keepdims=True)], axis=1)
            out[ok] = (bary * self.u[self._tri.simplices[s[ok]]]).sum(axis=1)
        return out


def mesh_region(outer: np.ndarray | None, hole: np.ndarray | None, lc: float):
    """Triangles of {inside ``outer``} minus {inside ``hole``} (either may be None: the unit square [-.5,.5]^2 stands
    in for a missing outer polygon).  Polygons are (m, 2) in (y, z).  Returns (points, tris, boundary node ids)."""
    from scipy.spatial import Delaunay, cKDTree
    square = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])
    ob = square if outer is None else outer
    bpts = [_resample_closed(ob, lc)]
    if hole is not None:
        bpts.append(_resample_closed(hole, lc))
    bpts = np.concatenate(bpts)
    lo, hi = ob.min(axis=0), ob.max(axis=0)
    ny = max(2, int(round((hi[1] - lo[1]) / (lc * np.sqrt(3.0) / 2.0))))
    nx = max(2, int(round((hi[0] - lo[0]) / lc)))
    dx, dy = (hi[0] - lo[0]) / nx, (hi[1] - lo[1]) / ny
    jj, ii = np.meshgrid(np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    lat = np.stack([lo[0] + (ii + 0.5 * (jj % 2)) * dx, lo[1] + jj * dy], axis=-1).reshape(-1, 2)
    inside = points_in_polygon(lat, ob)
    if hole is not None:
        inside &= ~points_in_polygon(lat, hole)
    lat = lat[inside]
    d, _ = cKDTree(bpts).query(lat)
    lat = lat[d > 0.6 * lc]
    pts = np.concatenate([bpts, lat])
    tris = Delaunay(pts).simplices
    cen = pts[tris].mean(axis=1)
    keep = points_in_polygon(cen, ob)
    if hole is not None:
        keep &= ~points_in_polygon(cen, hole)
    a = pts[tris]
    area2 = np.abs((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    keep &= area2 > 1e-12 * lc * lc
    tris = tris[keep]
    return pts, tris.astype(np.int32), np.arange(len(bpts))


def solve_velocity_field(pts: np.ndarray, tris: np.ndarray, boundary: np.ndarray, p: float = 10.0):
    """solve_velocity_field (:240-291): (u, area, average) of -Lap u = p, u = 0 on the boundary nodes, P1."""
    X = pts[tris]
    J = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]], axis=2)
    det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
    area_t = 0.5 * np.abs(det)
    K = np.linalg.inv(J)
    g = np.concatenate([-K.sum(axis=1, keepdims=True), K], axis=1)          # (e, 3, 2)
    Ke = area_t[:, None, None] * np.einsum("eai,ebi->eab", g, g)
    n = len(pts)
    A = sp.coo_matrix((Ke.ravel(), (np.repeat(tris, 3, axis=1).ravel(), np.tile(tris, (1, 3)).ravel())), shape=(n, n)).tocsr()
    b = np.zeros(n)
    np.add.at(b, tris.ravel(), np.repeat(p * area_t / 3.0, 3))
    used = np.zeros(n, dtype=bool)
    used[tris.ravel()] = True
    free = used.copy()
    free[boundary] = False
    u = np.zeros(n)
    idx = np.nonzero(free)[0]
    u[idx] = spla.splu(sp.csc_matrix(A[idx][:, idx])).solve(b[idx])
    area = float(area_t.sum())
    avg = float(np.sum(area_t * u[tris].mean(axis=1))) / area
    return u, area, avg


@dataclass
class InletProfiles:
    """What solve_inlet_profiles (:294-353) returns, in evaluable form: stream 1 inside the inner contour, stream 2
    between the outer contour and the duct wall; the dark band between the two contours is the nozzle wall."""
    inner: RegionSolution
    outer: RegionSolution
    contour_inner: np.ndarray       # (m, 2) polygon vertices as (z, y) (the reference's column order)
    contour_outer: np.ndarray
    mesh_lc: tuple

    @property
    def area_1(self):
        return self.inner.area

    @property
    def area_2(self):
        return self.outer.area

    def profile_1(self, x):
        return self.inner(x[:, 1], x[:, 2])

    def profile_2(self, x):
        return self.outer(x[:, 1], x[:, 2])

    def region_at(self, y, z):
        """1 inside the inner contour, 2 between the outer contour and the duct wall, 0 in the band."""
        q = np.stack([np.asarray(y, float), np.asarray(z, float)], axis=1)
        r = np.zeros(len(q), dtype=np.int8)
        r[points_in_polygon(q, self.contour_inner[:, ::-1])] = 1
        r[~points_in_polygon(q, self.contour_outer[:, ::-1]) & (np.abs(q) <= 0.5 + 1e-12).all(axis=1)] = 2
        return r


def solve_inlet_profiles(img_fname: str, flowrate_ratio: float, *, max_pixels: int | None = None) -> InletProfiles:
    """image2inlet.solve_inlet_profiles (:294-353).  ``max_pixels`` box-filters a larger image down first (the
    reference works at full resolution; the contours move by less than a pixel of the reduced image)."""
    gray = load_image(img_fname)
    if max_pixels and max(gray.shape) > max_pixels:
        f = int(np.ceil(max(gray.shape) / max_pixels))
        h, w = (gray.shape[0] // f) * f, (gray.shape[1] // f) * f
        gray = gray[:h, :w].reshape(h // f, f, w // f, f).mean(axis=(1, 3))
    contours = get_contours(gray)
    if len(contours) != 2:
        raise ValueError(f"the inlet image must give exactly two contours (outer and inner edge of the nozzle wall), "
                         f"found {len(contours)}")                       # image2gmsh3D.py:531-533
    c_in, lc_in = optimize_contour(contours[1])                          # process_2_channel_mesh_model (:217-223)
    c_out, lc_out = optimize_contour(contours[0])
    sols = []
    for poly, lc, hole, q in ((c_in[:, ::-1], lc_in, False, flowrate_ratio), (c_out[:, ::-1], lc_out, True, 1.0 - flowrate_ratio)):
        if hole:
            pts, tris, bnd = mesh_region(None, poly, lc)
        else:
            pts, tris, bnd = mesh_region(poly, None, lc)
        u, area, avg = solve_velocity_field(pts, tris, bnd)
        u = u / avg                                                      # average = 1 (:323-324)
        u = u * (q / area)                                               # flow_u = ratio / area (:336-339)
        sols.append(RegionSolution(pts, tris, u, area, avg))
    return InletProfiles(sols[0], sols[1], c_in, c_out, (lc_in, lc_out))
