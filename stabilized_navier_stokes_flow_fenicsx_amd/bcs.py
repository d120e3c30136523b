"""Dirichlet boundary data for the node-blocked P1-P1 space (dof = 4*node + c,
c in {ux,uy,uz,p}).

Mirrors the reference's ``dirichletbc`` lists:
  * channel  [wall, inlet_1, inlet_2, outlet]  NavierStokesChannelFlow.py:127-147
  * duct     [wall, inlet, outlet]             DuctStokesFlow.py:156-183
  * cavity   [noslip, lid, p(0,0,0)=0]         LidDrivenNavierStokesFlow.py:57-77
``set_bc`` walks the list in order, so for a dof that sits in several entries
the LAST entry's value wins (:146); ``DirichletSet.flatten`` reproduces that.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Sequence

import numpy as np

from .mesh import TetMesh


@dataclass
class DirichletBC:
    """One ``dirichletbc`` entry: nodes, which components, and values."""
    nodes: np.ndarray                 # (m,) node ids
    comps: tuple                      # subset of (0,1,2,3)
    values: np.ndarray                # (m, len(comps))


class DirichletSet:
    def __init__(self, mesh: TetMesh, bcs: Sequence[DirichletBC]):
        self.mesh = mesh
        self.bcs = list(bcs)

    def flatten(self) -> tuple[np.ndarray, np.ndarray]:
        """(mask uint8[ndof], g float64[ndof]); later entries override earlier ones."""
        nd = self.mesh.num_dofs
        mask = np.zeros(nd, dtype=np.uint8)
        g = np.zeros(nd, dtype=np.float64)
        for bc in self.bcs:
            for k, c in enumerate(bc.comps):
                d = 4 * bc.nodes.astype(np.int64) + c
                mask[d] = 1
                g[d] = bc.values[:, k]
        return mask, g


def _vel_bc(nodes, fn_or_val, pts) -> DirichletBC:
    nodes = np.asarray(nodes, dtype=np.int64)
    if callable(fn_or_val):
        vals = np.asarray(fn_or_val(pts[nodes]), dtype=np.float64).reshape(len(nodes), 3)
    else:
        vals = np.broadcast_to(np.asarray(fn_or_val, dtype=np.float64), (len(nodes), 3)).copy()
    return DirichletBC(nodes, (0, 1, 2), vals)


def _p_bc(nodes, val=0.0) -> DirichletBC:
    nodes = np.asarray(nodes, dtype=np.int64)
    return DirichletBC(nodes, (3,), np.full((len(nodes), 1), float(val)))


def duct_bcs(mesh: TetMesh, inlet_velocity=(1.0, 0.0, 0.0)) -> DirichletSet:
    """[wall u=0, inlet u=(1,0,0), outlet p=0]  (DuctStokesFlow.py:156-183)."""
    t = mesh.meta["tags"]
    return DirichletSet(mesh, [
        _vel_bc(mesh.facet_nodes(t["wall"]), (0.0, 0.0, 0.0), mesh.points),
        _vel_bc(mesh.facet_nodes(t["inlet"]), inlet_velocity, mesh.points),
        _p_bc(mesh.facet_nodes(t["outlet"])),
    ])


def channel_bcs(mesh: TetMesh, profile_1: Callable, profile_2: Callable) -> DirichletSet:
    """[wall, inlet_1, inlet_2, outlet] (NavierStokesChannelFlow.py:127-147).

    ``profile_k(points)->(m,)`` is the x-velocity of stream k; the reference
    interpolates a scalar Poisson profile into component 0 only (:150-157).
    """
    t = mesh.meta["tags"]

    def vec(profile):
        return lambda x: np.stack([profile(x), np.zeros(len(x)), np.zeros(len(x))], axis=1)

    return DirichletSet(mesh, [
        _vel_bc(mesh.facet_nodes(t["wall"]), (0.0, 0.0, 0.0), mesh.points),
        _vel_bc(mesh.facet_nodes(t["inlet_1"]), vec(profile_1), mesh.points),
        _vel_bc(mesh.facet_nodes(t["inlet_2"]), vec(profile_2), mesh.points),
        _p_bc(mesh.facet_nodes(t["outlet"])),
    ])


def cavity_bcs(mesh: TetMesh, lid_velocity=(1.0, 0.0, 0.0)) -> DirichletSet:
    """[no-slip walls, lid u=(1,0,0), p=0 at the origin] (LidDrivenNavierStokesFlow.py:57-77)."""
    t = mesh.meta["tags"]
    origin = np.nonzero(np.all(np.isclose(mesh.points, 0.0), axis=1))[0]
    return DirichletSet(mesh, [
        _vel_bc(mesh.facet_nodes(t["wall"]), (0.0, 0.0, 0.0), mesh.points),
        _vel_bc(mesh.facet_nodes(t["lid"]), lid_velocity, mesh.points),
        _p_bc(origin),
    ])


def two_stream_profiles(flowrate_ratio: float, inner_half_width: float = 0.25):
    """Analytic stand-ins for image2inlet.solve_inlet_profiles (:294-353).

    The reference solves -Lap u = 10 on each 2-D inlet region, normalises to
    mean 1 and scales by ratio/area resp. (1-ratio)/area (:323-339).  Without
    gmsh the synthetic channel uses separable bubble profiles with the same
    normalisation (unit mean on their region, then the same scaling).
    """
    a = inner_half_width
    area_1 = (2 * a) ** 2
    area_2 = 1.0 - area_1

    def p1(x):
        s = np.clip(1 - (x[:, 1] / a) ** 2, 0, None) * np.clip(1 - (x[:, 2] / a) ** 2, 0, None)
        return s * (9.0 / 4.0) * flowrate_ratio / area_1     # mean of (1-s^2)(1-t^2) on [-1,1]^2 = 4/9

    def p2(x):
        d_out = np.minimum(0.5 - np.abs(x[:, 1]), 0.5 - np.abs(x[:, 2]))
        d_in = np.maximum(np.abs(x[:, 1]), np.abs(x[:, 2])) - a
        w = 0.5 - a
        s = np.clip(4 * d_out * d_in / (w * w), 0, None)
        return s * 1.5 * (1 - flowrate_ratio) / area_2

    return p1, p2


def dfg_bcs(mesh: TetMesh, u_max: float = 0.45) -> DirichletSet:
    """[inflow, walls, obstacle] of DFG_3D_Validation.py:100-141: inlet u_x = 16 u_max y z (H-y)(H-z) / H^4, no slip
    on the channel walls and the pillar, NO pressure condition (the script defines one at the outlet but leaves it
    out of the list it passes on, :141) -- the outlet is the natural boundary."""
    t = mesh.meta["tags"]
    H = mesh.meta["width"]

    def inflow(x):
        ux = u_max * (4 * x[:, 1] * (H - x[:, 1]) / H ** 2) * (4 * x[:, 2] * (H - x[:, 2]) / H ** 2)
        return np.stack([ux, np.zeros(len(x)), np.zeros(len(x))], axis=1)

    return DirichletSet(mesh, [
        _vel_bc(mesh.facet_nodes(t["inlet"]), inflow, mesh.points),
        _vel_bc(mesh.facet_nodes(t["wall"]), (0.0, 0.0, 0.0), mesh.points),
        _vel_bc(mesh.facet_nodes(t["obstacle"]), (0.0, 0.0, 0.0), mesh.points),
    ])
