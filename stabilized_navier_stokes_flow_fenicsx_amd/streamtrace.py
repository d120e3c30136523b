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
