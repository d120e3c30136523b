"""Inlet image -> two-stream inlet data (host side; numpy / scipy / PIL only).

``channel_from_image`` drives the structured 4x1x1 channel with the inlet profiles of ``inlet_contours``
(the reference's own pipeline: contours -> FFT low-pass -> RDP -> P1 Poisson on a triangulation of each
region); ``method="pixel"`` selects the older pixel-grid restatement below, which the tests keep as an
independent sanity bound for the contour pipeline.

Pixel-grid variant:

The reference turns a black-on-white PNG of the nozzle wall (e.g.
``InletImages/PlusF_final.png``: a black band on white) into
  * two 2-D regions -- inside the band (stream 1) and outside it (stream 2) --
    meshed with gmsh (image2inlet.py:58-232),
  * a Poisson profile ``-Lap u = 10``, ``u = 0`` on the walls, per region (:240-291),
    normalised to mean 1 and scaled to ``ratio/area`` resp. ``(1-ratio)/area`` (:323-339),
  * a 3-D channel whose nozzle walls continue the band for ``x in [0, 0.5]``
    (image2gmsh3D.py:193-194).
skimage / rdp / shapely / gmsh are not available offline, so this
module does the same on the PIXEL grid: regions by connected components of the
non-wall pixels, the Poisson problem by 5-point finite differences on the region's
pixels (sparse LU), the same normalisation and scaling, and -- for the structured box
channel -- the nozzle walls as no-slip nodes inside the band for ``x <= nozzle_length``.
Coordinates follow get_contours (:86-91): y = (col - W/2)/W, z = -(row - H/2)/H.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.ndimage as ndi
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import mesh as M
from .bcs import DirichletBC, DirichletSet

MAX_PIXELS = 512          # larger images are box-filtered down before the 2-D solves


def load_image(path: str) -> np.ndarray:
    """Grayscale in [0,1] like load_image (:42-56): RGBA is composited over white, then luma."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("RGBA", "LA", "P"):
        im = im.convert("RGBA")
        bg = Image.new("RGBA", im.size, (255, 255, 255, 255))
        im = Image.alpha_composite(bg, im).convert("RGB")
    if im.mode != "L":
        rgb = np.asarray(im.convert("RGB"), dtype=np.float64) / 255.0
        g = 0.2125 * rgb[..., 0] + 0.7154 * rgb[..., 1] + 0.0721 * rgb[..., 2]      # skimage rgb2gray weights
    else:
        g = np.asarray(im, dtype=np.float64) / 255.0
    if max(g.shape) > MAX_PIXELS:
        from PIL import Image as I2
        s = MAX_PIXELS / max(g.shape)
        new = (max(8, int(round(g.shape[1] * s))), max(8, int(round(g.shape[0] * s))))
        g = np.asarray(I2.fromarray((g * 255).astype(np.uint8)).resize(new, I2.BOX), dtype=np.float64) / 255.0
    return g


@dataclass
class InletData:
    gray: np.ndarray
    region: np.ndarray          # per pixel: 0 wall, 1 inner stream, 2 outer stream
    u1: np.ndarray              # scaled profiles on the pixel grid (0 outside their region)
    u2: np.ndarray
    area_1: float
    area_2: float

    def _pix(self, y, z):
        H, W = self.gray.shape
        c = np.clip((np.asarray(y) + 0.5) * W - 0.5, 0, W - 1)
        r = np.clip((0.5 - np.asarray(z)) * H - 0.5, 0, H - 1)
        return r, c

    def region_at(self, y, z):
        r, c = self._pix(y, z)
        return self.region[np.round(r).astype(int), np.round(c).astype(int)]

    def _bilinear(self, f, y, z):
        r, c = self._pix(y, z)
        return ndi.map_coordinates(f, [r, c], order=1, mode="nearest")

    def profile_1(self, x):
        return self._bilinear(self.u1, x[:, 1], x[:, 2])

    def profile_2(self, x):
        return self._bilinear(self.u2, x[:, 1], x[:, 2])


def _poisson_on_mask(mask: np.ndarray, h: float, rhs: float = 10.0) -> np.ndarray:
    """-Lap u = rhs on the pixels of ``mask`` with u = 0 on every neighbouring pixel outside it
    (the wall band, and the duct wall beyond the image border)."""
    idx = -np.ones(mask.shape, dtype=np.int64)
    n = int(mask.sum())
    idx[mask] = np.arange(n)
    rows, cols, vals = [np.arange(n)], [np.arange(n)], [np.full(n, 4.0)]
    rr, cc = np.nonzero(mask)
    for dr, dc in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        r2, c2 = rr + dr, cc + dc
        ok = (r2 >= 0) & (r2 < mask.shape[0]) & (c2 >= 0) & (c2 < mask.shape[1])
        nb = np.full(n, -1, dtype=np.int64)
        nb[ok] = idx[r2[ok], c2[ok]]
        sel = nb >= 0
        rows.append(np.arange(n)[sel]); cols.append(nb[sel]); vals.append(np.full(int(sel.sum()), -1.0))
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    u = spla.splu(A).solve(np.full(n, rhs * h * h))
    out = np.zeros(mask.shape)
    out[mask] = u
    return out


def solve_inlet_profiles(img_fname: str, flowrate_ratio: float) -> InletData:
    """Counterpart of image2inlet.solve_inlet_profiles (:294-353) on the pixel grid."""
    gray = load_image(img_fname)
    H, W = gray.shape
    wall = gray < 0.5                                     # find_contours(gray, 0.5) (:61)
    lab, nlab = ndi.label(~wall)
    if nlab < 2:
        raise ValueError("the inlet image must separate an inner region from the outer one (a closed dark band)")
    border = np.unique(np.concatenate([lab[0], lab[-1], lab[:, 0], lab[:, -1]]))
    border = border[border > 0]
    outer = np.isin(lab, border)
    sizes = ndi.sum(np.ones_like(lab), lab, index=np.arange(1, nlab + 1))
    inner = np.zeros_like(outer)
    for k in range(1, nlab + 1):                          # area filter >= 5 % of the image (:76)
        if k not in border and sizes[k - 1] >= 0.05 * H * W:
            inner |= lab == k
    if not inner.any():
        raise ValueError("no enclosed region of at least 5 % of the image found")
    region = np.zeros(gray.shape, dtype=np.int8)
    region[inner] = 1
    region[outer] = 2
    h = 1.0 / W
    px = (1.0 / W) * (1.0 / H)
    out = []
    for mask, q in ((inner, flowrate_ratio), (outer, 1.0 - flowrate_ratio)):
        u = _poisson_on_mask(mask, h)
        area = float(mask.sum()) * px
        mean = float(u.sum()) * px / area
        u = u / mean                                      # average = 1 (:323-324)
        out.append((u * (q / area), area))               # flow_u = ratio / area (:336-339)
    return InletData(gray, region, out[0][0], out[1][0], out[0][1], out[1][1])


def channel_from_image(img_fname: str, flowrate_ratio: float, cells, *, nozzle_length: float = 0.5,
                       method: str = "contours"):
    """(mesh, DirichletSet, inlet data) of the 4x1x1 channel driven by an inlet image.

    Facet tags as image2gmsh3D.py:435-438 (inlet_1=1, inlet_2=2, outlet=3, wall=4): inlet facets are
    classified by the image region under their centroid, facets under the dark band are wall.  The
    nozzle walls (band extruded over x in [0, nozzle_length]) become no-slip nodes."""
    if method == "contours":
        from .inlet_contours import solve_inlet_profiles as solve_contours
        data = solve_contours(img_fname, flowrate_ratio, max_pixels=1024)
    elif method == "pixel":
        data = solve_inlet_profiles(img_fname, flowrate_ratio)
    else:
        raise ValueError("method must be 'contours' or 'pixel'")
    m = M.channel_mesh(cells)
    t = m.meta["tags"]
    inl = np.nonzero((m.facet_tags == t["inlet_1"]) | (m.facet_tags == t["inlet_2"]))[0]
    cen = m.points[m.facets[inl]].mean(axis=1)
    reg = data.region_at(cen[:, 1], cen[:, 2])
    m.facet_tags[inl[reg == 1]] = t["inlet_1"]
    m.facet_tags[inl[reg == 2]] = t["inlet_2"]
    m.facet_tags[inl[reg == 0]] = t["wall"]
    m.meta["kind"] = "channel-image"

    def vec(profile):
        return lambda x: np.stack([profile(x), np.zeros(len(x)), np.zeros(len(x))], axis=1)

    pts = m.points
    wall_nodes = m.facet_nodes(t["wall"])
    band = np.nonzero((pts[:, 0] <= nozzle_length + 1e-12) & (data.region_at(pts[:, 1], pts[:, 2]) == 0))[0]
    wall_nodes = np.union1d(wall_nodes, band)

    def bc(nodes, comps, vals):
        return DirichletBC(np.asarray(nodes, dtype=np.int64), comps, np.asarray(vals, dtype=np.float64))

    n1, n2 = m.facet_nodes(t["inlet_1"]), m.facet_nodes(t["inlet_2"])
    out = m.facet_nodes(t["outlet"])
    bcs = DirichletSet(m, [                                   # [wall, inlet_1, inlet_2, outlet] (:146)
        bc(wall_nodes, (0, 1, 2), np.zeros((len(wall_nodes), 3))),
        bc(n1, (0, 1, 2), vec(data.profile_1)(pts[n1])),
        bc(n2, (0, 1, 2), vec(data.profile_2)(pts[n2])),
        bc(out, (3,), np.zeros((len(out), 1))),
    ])
    # band nodes ON the inlet plane must stay no-slip although they may touch inlet facets
    mask, g = bcs.flatten()
    on_band_inlet = band[np.isclose(pts[band, 0], 0.0)]
    for c in range(3):
        g[4 * on_band_inlet + c] = 0.0
        mask[4 * on_band_inlet + c] = 1
    return m, (mask, g), data
