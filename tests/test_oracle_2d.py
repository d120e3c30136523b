"""2-D UGN oracle (oracle/forms2d.py) and the 2-D host code, on the CPU.

This is the part of the oracle the REFERENCE pins: Validation_Flow/DFG_2D_Validation.py:202-203 holds
C_d = 5.57953523384 / C_l = 0.010618948146 for the form of :141-163 with the functional of :195-200; the GPU test
`test_gpu_2d.py::test_dfg2d_series_converges_to_the_reference_constants` closes that loop at scale.  Here: the
literal restatement against its committed golden vectors, finite differences, and known answers (plane Poiseuille
flow, hydrostatic load on the obstacle)."""
import numpy as np
import pytest
import torch

from conftest import golden, rel
from oracle import forms2d as F2
from stabilized_navier_stokes_flow_fenicsx_amd import mesh2d as M2


def test_literal_forms_regenerate_golden():
    g = golden("ugn2d_elements.npz")
    for i in range(len(g["X"])):
        X, w, nu = torch.as_tensor(g["X"][i]), torch.as_tensor(g["W"][i].reshape(9)), float(g["nu"][i])
        assert rel(F2.ugn_residual_one(X, w, nu).numpy(), g["F"][i]) < 1e-14 or np.abs(g["F"][i]).max() < 1e-300
        J = torch.autograd.functional.jacobian(lambda ww: F2.ugn_residual_one(X, ww, nu), w).numpy()
        assert rel(J, g["J"][i]) < 1e-13
        assert rel(F2.stokes_matrix_one(X, 1.0, 0.2).numpy(), g["A_dfg"][i]) < 1e-14
    # the batched (vmap) path used for global assembly is the same function
    pts = g["X"].reshape(-1, 2)
    tris = np.arange(len(pts)).reshape(-1, 3)
    w4 = np.zeros((len(pts), 4))
    w4[:, [0, 1, 3]] = g["W"].reshape(-1, 3)
    for i in (0, 2, 5):
        R, J = F2.ugn_elements(pts, tris[i:i + 1], w4.ravel(), float(g["nu"][i]))
        assert rel(R[0], g["F"][i]) < 1e-14 and rel(J[0], g["J"][i]) < 1e-13


def test_golden_covers_both_branches_of_both_conditionals():
    """conditional(le(u_norm, 1e-8), ...) and conditional(le(Re_UGN, 3), ...) (:153,:158)"""
    g = golden("ugn2d_elements.npz")
    slow = re_lo = re_hi = 0
    for X, W, nu in zip(g["X"], g["W"], g["nu"]):
        e = np.array([X[0] - X[1], X[0] - X[2], X[1] - X[2]])
        h = np.sqrt((e * e).sum(1)).max()
        for q in F2.QPTS:
            phi = np.array([1 - q[0] - q[1], q[0], q[1]])
            un = np.linalg.norm(phi @ W[:, :2])
            slow += un <= 1e-8
            re_lo += un * h / (2 * nu) <= 3
            re_hi += un * h / (2 * nu) > 3
    assert slow >= 3 and re_lo >= 6 and re_hi >= 6


@pytest.mark.parametrize("nu", [1e-3, 0.3])
def test_jacobian_is_derivative_of_residual(nu):
    rng = np.random.default_rng(9)
    X = torch.as_tensor(rng.normal(size=(6, 3, 2)) * 0.1)
    w = rng.normal(size=(6, 9))
    f = torch.func.vmap(lambda X, w: F2.ugn_residual_one(X, w, nu))
    J = torch.func.vmap(torch.func.jacrev(lambda X, w: F2.ugn_residual_one(X, w, nu), argnums=1))(X, torch.as_tensor(w)).numpy()
    eps = 1e-6
    for k in range(9):
        d = np.zeros(9); d[k] = eps
        fd = (f(X, torch.as_tensor(w + d)).numpy() - f(X, torch.as_tensor(w - d)).numpy()) / (2 * eps)
        assert np.abs(fd - J[:, :, k]).max() < 2e-7 * max(1.0, np.abs(J).max())


def test_form_is_invariant_under_cell_local_vertex_order():
    """Unlike the 3-D G-metric form, the UGN form only sees h = CellDiameter and a symmetric rule: any local
    renumbering of a triangle permutes the element vector and nothing else."""
    g = golden("ugn2d_elements.npz")
    X, W, nu = g["X"][2], g["W"][2], float(g["nu"][2])
    F0 = F2.ugn_residual_one(torch.as_tensor(X), torch.as_tensor(W.reshape(9)), nu).numpy().reshape(3, 3)
    for perm in ([1, 2, 0], [0, 2, 1]):
        Fp = F2.ugn_residual_one(torch.as_tensor(X[perm]), torch.as_tensor(W[perm].reshape(9)), nu).numpy().reshape(3, 3)
        assert rel(Fp, F0[perm]) < 1e-13


def test_cavity_golden_newton_reproduces():
    g = golden("cavity2d_8.npz")
    nu = 1.0 / float(g["Re"])
    U = F2.solve_stokes2d(g["points"], g["tris"], g["mask"], g["g"], nu, (1.0 / 3.0) / (4 * nu))
    assert rel(U, g["U_stokes"]) < 1e-11
    w, info = F2.newton2d(g["points"], g["tris"], U, nu, g["mask"], g["g"])
    assert info["converged"] and info["its"] == int(g["its"]) and rel(w, g["w_newton"]) < 1e-10
    W = w.reshape(-1, 4)
    assert np.all(W[:, 2] == 0.0)
    top = np.isclose(g["points"][:, 1], 1.0)
    assert np.allclose(W[top, 0], 1.0) and np.allclose(W[top, 1], 0.0)       # lid wins on the corners (:77)


def test_plane_poiseuille_flow_is_recovered_under_refinement():
    """Known answer for the whole 2-D stack: between two plates the parabola u = 4 U y (H-y)/H^2 with
    p = 8 nu U (L - x)/H^2 solves the NS equations exactly, satisfies the natural outflow condition of the
    (grad u, grad v) form; for P1 the viscous part of the strong residual vanishes (div(sym(grad u)) = 0), so the
    stabilisation is consistent only to O(tau |grad p|) and the observed order is ~1.4 (0.054 -> 0.021)."""
    nu, U0, L, H = 0.05, 1.0, 2.0, 1.0
    errs = []
    for n in (6, 12):
        m = M2.rectangle_mesh(2 * n, n, (0.0, 0.0), (L, H))
        x, y = m.points[:, 0], m.points[:, 1]
        mask = np.zeros(m.num_dofs, np.uint8)
        g = np.zeros(m.num_dofs)
        wall = np.isclose(y, 0) | np.isclose(y, H)
        inlet = np.isclose(x, 0)
        for nodes, val in ((np.nonzero(inlet)[0], 4 * U0 * y[inlet] * (H - y[inlet]) / H ** 2), (np.nonzero(wall)[0], 0.0)):
            mask[4 * nodes] = mask[4 * nodes + 1] = 1
            g[4 * nodes] = val
            g[4 * nodes + 1] = 0.0
        w0 = F2.solve_stokes2d(m.points, m.tris, mask, g, nu, 0.2)
        w, info = F2.newton2d(m.points, m.tris, w0, nu, mask, g)
        assert info["converged"]
        W = w.reshape(-1, 4)
        ue = 4 * U0 * y * (H - y) / H ** 2
        errs.append(np.sqrt(np.mean((W[:, 0] - ue) ** 2 + W[:, 1] ** 2)))
        pe = 8 * nu * U0 * (L - x) / H ** 2
        assert np.abs(W[:, 3] - pe).max() < (0.25 if n == 6 else 0.1) * pe.max()
    assert errs[1] < 0.5 * errs[0] and errs[1] < 0.03


def test_drag_lift_functional_known_answers_and_oracle_loops():
    g = golden("dfg2d_level05.npz")
    m = M2.TriMesh(g["points"], g["tris"], g["facets"], g["facet_tags"], meta={"tags": dict(M2.DFG2D_TAGS)})
    cd, cl = M2.drag_lift_2d(m, g["w_newton"], 1e-3)
    assert abs(cd - float(g["cd"])) < 1e-12 * abs(cd) and abs(cl - float(g["cl"])) < 1e-10 * abs(cl)
    assert 5.0 < cd < 5.6 and 0.005 < cl < 0.015                  # coarse mesh, on its way to 5.5795 / 0.010619
    # hydrostatic load: u = 0, p = a x + b y  =>  oint p n ds = grad p * area of the (polygonal) obstacle
    a, b = 0.7, -1.3
    w = np.zeros(m.num_dofs)
    w[3::4] = a * m.points[:, 0] + b * m.points[:, 1]
    ob = m.facets[m.find(M2.DFG2D_TAGS["obstacle"])]
    nodes = np.unique(ob)
    c = m.points[nodes].mean(axis=0)
    ang = np.argsort(np.arctan2(m.points[nodes, 1] - c[1], m.points[nodes, 0] - c[0]))
    P = m.points[nodes][ang]
    area = 0.5 * abs(np.sum(P[:, 0] * np.roll(P[:, 1], -1) - np.roll(P[:, 0], -1) * P[:, 1]))
    cd, cl = M2.drag_lift_2d(m, w, 1e-3)
    s = 2.0 / (0.2 ** 2 * 0.1)
    assert abs(cd - (-s * a * area)) < 1e-10 and abs(cl - (-s * b * area)) < 1e-10
    cdo, clo = F2.drag_lift_loops(m.points, m.tris, ob, w, 1e-3)
    assert abs(cd - cdo) < 1e-10 and abs(cl - clo) < 1e-10


def test_dfg2d_mesh_geometry_tags_and_msh_round_trip(tmp_path):
    m = M2.dfg_2d_mesh(1.0)
    a = m.points[m.tris]
    area = 0.5 * ((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    assert np.all(area > 0)                                                   # counter-clockwise, no inverted cells
    assert abs(area.sum() - (2.2 * 0.41 - np.pi * 0.05 ** 2)) < 2e-5          # dfg_pillar_2D.geo:4-9
    assert M2.triangle_quality(m).min() > 0.6
    ob = m.facet_nodes(M2.DFG2D_TAGS["obstacle"])
    assert np.allclose(np.linalg.norm(m.points[ob] - [0.2, 0.2], axis=1), 0.05, atol=1e-12)
    assert np.allclose(m.points[m.facet_nodes(M2.DFG2D_TAGS["inlet"]), 0], 0.0)
    assert np.allclose(m.points[m.facet_nodes(M2.DFG2D_TAGS["outlet"]), 0], 2.2)
    mask, g = M2.dfg2d_bcs(m).flatten()
    assert mask[3::4].sum() == 0                                             # no pressure condition (:90)
    assert abs(g[0::4].max() - 0.3) < 2e-3                                     # 4 * 0.3 * y (H - y) / H^2 (:52)
    f = str(tmp_path / "dfg.msh")
    M2.write_msh2_2d(m, f)
    r = M2.read_msh_2d(f, reorder=False)
    assert np.array_equal(r.tris, m.tris) and np.allclose(r.points, m.points, atol=0, rtol=0)
    assert sorted(map(tuple, np.sort(r.facets, 1).tolist())) == sorted(map(tuple, np.sort(m.facets, 1).tolist()))
    assert np.bincount(r.facet_tags).tolist() == np.bincount(m.facet_tags).tolist()
    c = M2.rectangle_mesh(4)
    assert c.tris[:2].tolist() == [[0, 1, 6], [0, 5, 6]]                      # create_rectangle, diagonal "right"


def test_read_msh41_2d_with_physical_curves(tmp_path):
    """gmsh 4.1 ASCII as ``gmsh dfg_pillar_2D.geo -2 -format msh4`` writes it: physical groups arrive through the
    entity table ($Entities: curve -> physical tag), elements are grouped per entity (DFG_2D_Validation.py:28)."""
    txt = """$MeshFormat
4.1 0 8
$EndMeshFormat
$PhysicalNames
3
1 2 "inlet"
1 4 "walls"
2 1 "fluid"
$EndPhysicalNames
$Entities
4 4 1 0
1 0 0 0 0
2 1 0 0 0
3 1 1 0 0
4 0 1 0 0
1 0 0 0 1 0 0 1 4 2 1 -2
2 1 0 0 1 1 0 1 4 2 2 -3
3 0 1 0 1 1 0 1 4 2 3 -4
4 0 0 0 0 1 0 1 2 2 4 -1
1 0 0 0 1 1 0 1 1 4 1 2 3 4
$EndEntities
$Nodes
2 5 1 5
0 1 0 4
1
2
3
4
0 0 0
1 0 0
1 1 0
0 1 0
2 1 0 1
5
0.5 0.5 0
$EndNodes
$Elements
5 8 1 8
1 1 1 1
1 1 2
1 2 1 1
2 2 3
1 3 1 1
3 3 4
1 4 1 1
4 4 1
2 1 2 4
5 1 2 5
6 2 3 5
7 3 4 5
8 4 1 5
$EndElements
"""
    f = tmp_path / "sq.msh"
    f.write_text(txt)
    m = M2.read_msh_2d(str(f), reorder=False)
    assert m.num_nodes == 5 and m.num_cells == 4 and m.points.shape == (5, 2)
    assert sorted(m.facet_tags.tolist()) == [2, 4, 4, 4]                       # curve 4 carries "inlet" (2), the rest "walls"
    inlet = m.facets[m.find(2)][0]
    assert set(inlet.tolist()) == {3, 0}                                       # nodes 4-1 of the file, zero-based
    a = m.points[m.tris]
    area = 0.5 * np.abs((a[:, 1, 0] - a[:, 0, 0]) * (a[:, 2, 1] - a[:, 0, 1]) - (a[:, 2, 0] - a[:, 0, 0]) * (a[:, 1, 1] - a[:, 0, 1]))
    assert area.sum() == pytest.approx(1.0)
