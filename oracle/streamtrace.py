"""CPU oracle of the particle tracer: the reference's own recipe, scipy solve_ivp(RK45) per seed
(NavierStokes/streamtrace.py:208-232, :357-383) with a brute-force point location instead of the
dolfinx bounding-box tree.  Test infrastructure (see oracle/__init__.py)."""
from __future__ import annotations

import numpy as np
from scipy.integrate import solve_ivp


class P1Field:
    def __init__(self, points, tets, vel):
        self.X = points[tets]                                     # (E,4,3)
        self.tets = tets
        self.vel = np.asarray(vel).reshape(-1, 3)
        T = np.stack([self.X[:, 1] - self.X[:, 0], self.X[:, 2] - self.X[:, 0], self.X[:, 3] - self.X[:, 0]], axis=2)
        self.Tinv = np.linalg.inv(T)
        self.lo, self.hi = self.X.min(axis=1), self.X.max(axis=1)

    def __call__(self, x):
        cand = np.nonzero(np.all((self.lo <= x + 1e-12) & (self.hi >= x - 1e-12), axis=1))[0]
        if len(cand) == 0:
            return np.zeros(3)
        lam = np.einsum("eij,ej->ei", self.Tinv[cand], x - self.X[cand, 0])
        lam = np.concatenate([1 - lam.sum(axis=1, keepdims=True), lam], axis=1)
        ok = np.nonzero(lam.min(axis=1) >= -1e-12)[0]
        if len(ok) == 0:
            return np.zeros(3)                                    # outside the mesh: zero velocity (:149-153)
        e = cand[ok[0]]
        return lam[ok[0]] @ self.vel[self.tets[e]]


def trace(field: P1Field, seed, reverse=False, x_stop=None, t_end=20.0, max_step=0.125, rtol=1e-3, atol=1e-6):
    sgn = -1.0 if reverse else 1.0
    x_stop = (0.13 if reverse else 3.7) if x_stop is None else x_stop
    fun = lambda t, y: sgn * field(y)
    ev_speed = lambda t, y: np.linalg.norm(fun(t, y)) - 1e-6
    ev_speed.terminal, ev_speed.direction = True, -1
    ev_plane = lambda t, y: y[0] - x_stop
    ev_plane.terminal, ev_plane.direction = True, (-1 if reverse else 1)
    sol = solve_ivp(fun, (0, t_end), np.asarray(seed, float), method="RK45", events=(ev_speed, ev_plane),
                    max_step=max_step, rtol=rtol, atol=atol)
    status = 0
    if sol.status == 1:
        status = 1 if len(sol.t_events[0]) else 2
    return sol.y[:, -1], sol.t[-1], status, len(sol.t) - 1
