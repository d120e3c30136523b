"""Particle tracing through the computed P1 velocity field on the GPU.

Counterpart of NavierStokes/streamtrace.py: the reference traces every
seed with its own ``scipy.integrate.solve_ivp(RK45, max_step=0.125, t in [0, 20])`` call, a
bounding-box-tree lookup per velocity evaluation, a thread pool for the forward pass (:208-250) and
an MPI task farm for the reverse pass (:385-446).  Here all seeds are traced by one HIP kernel
(``sns_streamtrace``, csrc/sns_trace.hip: same RK45, controller, events); the host only builds the
tet face adjacency and locates the seeds' starting tets.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .interpolate import locate_points
from .mesh import TetMesh

X_STOP_FORWARD = 3.7      # position_event            streamtrace.py:180-183
X_STOP_REVERSE = 0.13     # reverse_position_event    :185-188
STATUS = {0: "t_end", 1: "stopped (speed < 1e-6 / left the mesh)", 2: "reached the stop plane", 3: "step underflow"}


def tet_face_neighbors(tets: np.ndarray) -> np.ndarray:
    """nbr[t, a] = tet sharing the face opposite local vertex a of tet t, -1 on the boundary."""
    E = len(tets)
    faces = np.concatenate([tets[:, [1, 2, 3]], tets[:, [0, 2, 3]], tets[:, [0, 1, 3]], tets[:, [0, 1, 2]]]).astype(np.int64)
    fs = np.sort(faces, axis=1)
    n = int(tets.max()) + 1
    key = (fs[:, 0] * n + fs[:, 1]) * n + fs[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    same = ks[1:] == ks[:-1]
    nbr = -np.ones(4 * E, dtype=np.int32)
    i0, i1 = order[:-1][same], order[1:][same]                   # paired faces (each interior face appears twice)
    nbr[i0] = (i1 % E).astype(np.int32)
    nbr[i1] = (i0 % E).astype(np.int32)
    return np.ascontiguousarray(nbr.reshape(4, E).T)             # face block a <-> opposite vertex a


def run_streamtrace(mesh: TetMesh, velocity, seeds, *, reverse: bool = False, x_stop: float | None = None,
                    t_end: float = 20.0, max_step: float = 0.125, rtol: float = 1e-3, atol: float = 1e-6,
                    speed_min: float = 1e-6, device="cuda:0", nbr=None):
    """Trace ``seeds`` (m,3) through ``velocity`` (n,3); returns dict(pos, t, status, steps) (numpy)."""
    import torch
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("run_streamtrace needs a HIP device; there is no CPU fallback")
    dev = torch.device(device)
    seeds = np.ascontiguousarray(seeds, dtype=np.float64).reshape(-1, 3)
    vel = np.ascontiguousarray(np.asarray(velocity, dtype=np.float64).reshape(-1, 3))
    if vel.shape[0] != mesh.num_nodes:
        raise ValueError("velocity must have one row per mesh node")
    if x_stop is None:
        x_stop = X_STOP_REVERSE if reverse else X_STOP_FORWARD
    if nbr is None:
        nbr = tet_face_neighbors(mesh.tets)
    seed_tet, _ = locate_points(mesh, seeds) if len(seeds) else (np.zeros(0, np.int64), None)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    d_pts, d_tets = t(mesh.points, torch.float64), t(mesh.tets, torch.int32)
    d_nbr, d_vel = t(nbr, torch.int32), t(vel, torch.float64)
    d_seeds, d_st = t(seeds, torch.float64), t(seed_tet.astype(np.int32), torch.int32)
    m = len(seeds)
    pos = torch.empty((m, 3), dtype=torch.float64, device=dev)
    tt = torch.empty(m, dtype=torch.float64, device=dev)
    status = torch.empty(m, dtype=torch.int32, device=dev)
    steps = torch.empty(m, dtype=torch.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.sns_streamtrace(mesh.num_nodes, mesh.num_tets, p(d_pts), p(d_tets), p(d_nbr), p(d_vel), m,
                                       p(d_seeds), p(d_st), 1 if reverse else 0, t_end, max_step, rtol, atol,
                                       float(x_stop), speed_min, p(pos), p(tt), p(status), p(steps), stream))
    return dict(pos=pos.cpu().numpy(), t=tt.cpu().numpy(), status=status.cpu().numpy(), steps=steps.cpu().numpy())


def make_rev_streamtrace_seeds(minx, maxx, miny, maxy, numpoints, x_plane: float = 3.9):
    """N x N seeds on the plane x = 3.9 (streamtrace.py:346-355)."""
    a, b = np.meshgrid(np.linspace(minx, maxx, numpoints), np.linspace(miny, maxy, numpoints))
    pts = np.stack([a, b], axis=-1).reshape(-1, 2)
    return np.hstack([np.full((len(pts), 1), x_plane), pts])


def alpha_shape_exterior(points: np.ndarray, alpha: float = 0.2) -> np.ndarray:
    """Exterior ring (k, 2), closed, of the LARGEST polygon of the alpha shape of a 2-D point set -- what
    ``alphashape.alphashape(points, 0.2)`` + the largest-polygon pick of expand_streamtace (streamtrace.py:292-311)
    deliver: Delaunay triangles with circumradius < 1 / alpha, their union, its outer boundary."""
    from scipy.spatial import Delaunay
    pts = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 2), axis=0)
    if len(pts) < 4:
        raise ValueError("alpha shape needs at least 4 distinct points")
    tri = Delaunay(pts).simplices
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    la, lb, lc = np.linalg.norm(b - c, axis=1), np.linalg.norm(c - a, axis=1), np.linalg.norm(a - b, axis=1)
    area = 0.5 * np.abs((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1]))
    with np.errstate(divide="ignore", invalid="ignore"):
        radius = la * lb * lc / (4.0 * area)
    keep = tri[(area > 0) & (radius < 1.0 / alpha)] if alpha > 0 else tri[area > 0]
    if len(keep) == 0:
        raise ValueError("alpha shape is empty: degenerate point set or alpha too large")
    e = np.concatenate([keep[:, [0, 1]], keep[:, [1, 2]], keep[:, [2, 0]]])
    es = np.sort(e, axis=1)
    key = es[:, 0].astype(np.int64) * len(pts) + es[:, 1]
    uniq, cnt = np.unique(key, return_counts=True)
    bnd = es[np.isin(key, uniq[cnt == 1])]
    nbr = {}
    for i, j in bnd.tolist():
        nbr.setdefault(i, []).append(j)
        nbr.setdefault(j, []).append(i)
    rings, seen = [], set()
    for s in sorted(nbr):
        if s in seen:
            continue
        ring, prev, cur = [s], None, s
        seen.add(s)
        while True:
            cand = [n for n in nbr[cur] if n != prev and n not in seen]
            if not cand:
                break
            prev, cur = cur, cand[0]
            ring.append(cur)
            seen.add(cur)
        rings.append(np.array(ring + [ring[0]]))

    def ring_area(r):
        q = pts[r]
        return 0.5 * abs(np.sum(q[:-1, 0] * q[1:, 1] - q[1:, 0] * q[:-1, 1]))

    best = max(rings, key=ring_area)
    return pts[best]


def expand_streamtace(pointsy, pointsz, blurr: float = 0.2, alpha: float = 0.2):
    """(min y, max y, min z, max z) of the alpha shape of the forward trace's arrival points, its extreme vertices
    pushed by 20 % exactly as expand_streamtace (sic; streamtrace.py:292-343) does -- including the branch for
    extents that do not straddle zero."""
    ring = alpha_shape_exterior(np.stack([np.squeeze(pointsy), np.squeeze(pointsz)], axis=1), alpha)
    x, y = ring[:, 0].copy(), ring[:, 1].copy()
    for v in (x, y):
        if v.min() <= 0 and v.max() >= 0:
            i0 = int(np.argmin(v))
            v[i0] = -1 * abs(v[i0] * blurr) + -1 * abs(v[i0])
            i1 = int(np.argmax(v))
            v[i1] = v[i1] * blurr + v[i1]
        else:
            i0 = int(np.argmin(v))
            v[i0] = -1 * v[i0] * blurr + v[i0]
            i1 = int(np.argmax(v))
            v[i1] = v[i1] * blurr + v[i1]
    return float(x.min()), float(x.max()), float(y.min()), float(y.max())


def find_seed_end(rev_pointsy, rev_pointsz, seeds, contour):
    """Seeds (their (y, z) on the plane x = 3.9) whose reverse trace ends INSIDE the inner inlet contour
    (streamtrace.py:536-553: ``sk.measure.points_in_poly`` against contour[:, 1:3])."""
    from .inlet_contours import points_in_polygon
    q = np.stack([np.asarray(rev_pointsy, float), np.asarray(rev_pointsz, float)], axis=1)
    inside = points_in_polygon(q, np.asarray(contour, float)[:, 1:3])
    return np.asarray(seeds, float)[inside][:, 1:3]


def update_contour(img_fname: str, max_pixels: int | None = 1024):
    """Inner inlet contour as (m, 3) rows (0, y, z) (streamtrace.py:132-143)."""
    from . import inlet_contours as IC
    gray = IC.load_image(img_fname)
    if max_pixels and max(gray.shape) > max_pixels:
        f = int(np.ceil(max(gray.shape) / max_pixels))
        h, w = (gray.shape[0] // f) * f, (gray.shape[1] // f) * f
        gray = gray[:h, :w].reshape(h // f, f, w // f, f).mean(axis=(1, 3))
    contour, _ = IC.optimize_contour(IC.get_contours(gray)[1])
    return np.hstack([np.zeros((len(contour), 1)), contour[:, [1, 0]]])


def read_mesh_and_function(fname_base: str, function_name: str = "Velocity", function_dim: int = 3):
    """(TetMesh, nodal values) from ``<fname_base>.xdmf/.h5`` -- the hand-over between solver and post-processing
    (streamtrace.py:58-130 reads ``h5f["Function"][function_name]["0"]``)."""
    from .drivers import read_xdmf_function
    pts, cells, vals = read_xdmf_function(fname_base, function_name)
    mesh = TetMesh(np.ascontiguousarray(pts, dtype=np.float64), np.ascontiguousarray(cells, dtype=np.int32),
                   np.zeros((0, 3), np.int32), np.zeros(0, np.int32), name=fname_base)
    return mesh, np.ascontiguousarray(vals[:, :function_dim], dtype=np.float64)


def for_and_rev_streamtrace_files(num_seeds: int, img_fname: str, Re, Folder_name: str, *, out_dir: str | None = None,
                                  **kw):
    """The reference's for_and_rev_streamtrace(num_seeds, limits, img_fname, ..., Re, Folder_name) (:556-665) without
    the matplotlib figures: velocity re-read from ``{Folder}/Re{Re}ChannelVelocity.{xdmf,h5}`` (:590), inner inlet
    contour from the image (:598), forward seeds = nodes of the inner inlet mesh (:606), forward trace, alpha-shape
    bound + 20 % blur (:620), N x N reverse seeds on x = 3.9 (:623), reverse trace (:640), point-in-polygon filter
    (:644); ``rev_seeds.csv`` and ``final_output.csv`` as save_figs writes them (:519-520)."""
    import os
    from . import inlet_contours as IC
    mesh, vel = read_mesh_and_function(os.path.join(Folder_name, f"Re{Re}ChannelVelocity"), "Velocity", 3)
    contour = update_contour(img_fname)
    prof = IC.solve_inlet_profiles(img_fname, 0.5, max_pixels=1024)                  # inner_contour_mesh_func (:190-197)
    used = np.unique(prof.inner.tris)
    inner_mesh = np.hstack([np.zeros((len(used), 1)), prof.inner.points[used]])
    nbr = tet_face_neighbors(mesh.tets)
    fwd = run_streamtrace(mesh, vel, inner_mesh, reverse=False, nbr=nbr, **kw)
    ok = fwd["pos"][:, 0] > 0.5                                                     # streamtrace_pool (:214-222)
    if not ok.any():
        raise RuntimeError("no forward particle left the nozzle")
    miny, maxy, minz, maxz = expand_streamtace(fwd["pos"][ok, 1], fwd["pos"][ok, 2])
    seeds = make_rev_streamtrace_seeds(miny, maxy, minz, maxz, num_seeds)
    rev = run_streamtrace(mesh, vel, seeds, reverse=True, nbr=nbr, **kw)
    end = np.where((rev["pos"][:, 0] < 0.5)[:, None], rev["pos"], 10.0)             # reverse_streamtrace_pool (:372-383)
    final_output = find_seed_end(end[:, 1], end[:, 2], seeds, contour)
    if out_dir is not None:
        np.savetxt(os.path.join(out_dir, "rev_seeds.csv"), seeds, delimiter=",")
        np.savetxt(os.path.join(out_dir, "final_output.csv"), final_output, delimiter=",")
    return dict(forward=fwd, reverse=rev, rev_seeds=seeds, final_output=final_output, contour=contour,
                bounds=(miny, maxy, minz, maxz))


def for_and_rev_streamtrace(mesh: TetMesh, velocity, inner_points, num_seeds: int = 50, blur: float = 0.2, **kw):
    """Forward trace of the inner-stream inlet points to x = 3.7, a blurred bounding box of where they
    arrive, N x N reverse seeds on x = 3.9 and their reverse trace to x = 0.13 (the structure of
    for_and_rev_streamtrace, :556-640, without the plotting / alpha-shape post-processing)."""
    nbr = tet_face_neighbors(mesh.tets)
    fwd = run_streamtrace(mesh, velocity, inner_points, reverse=False, nbr=nbr, **kw)
    ok = (fwd["status"] == 2) & (fwd["pos"][:, 0] > 0.5)                          # :224-232
    arrived = fwd["pos"][ok]
    if len(arrived) == 0:
        raise RuntimeError("no forward particle reached the stop plane")
    lo, hi = arrived[:, 1:].min(axis=0), arrived[:, 1:].max(axis=0)
    lo, hi = lo - blur * np.abs(lo), hi + blur * np.abs(hi)                       # 20 % blur (:292-343)
    seeds = make_rev_streamtrace_seeds(lo[0], hi[0], lo[1], hi[1], num_seeds)
    rev = run_streamtrace(mesh, velocity, seeds, reverse=True, nbr=nbr, **kw)
    return dict(forward=fwd, reverse=rev, rev_seeds=seeds, arrived=arrived)
