"""Oracle pinning, element level (CPU).  The reference has no fixtures for this
path (parity unpinned, oracle/__init__.py), so the oracle is pinned by
 (1) the literal term-by-term restatement of the UFL text + autograd Jacobian,
 (2) finite differences of the residual, (3) structural known answers."""
import numpy as np
import pytest

from conftest import golden, rel
from oracle import element as el, forms_literal as fl


def test_closed_form_matches_golden_literal_forms():
    g = golden("element_ns.npz")
    R, J = el.ns_element(g["X"], g["W"], 1.0, want_jac=False)  # dummy call shape check
    for i in range(len(g["X"])):
        R, J = el.ns_element(g["X"][i:i + 1], g["W"][i:i + 1], float(g["Re"][i]))
        assert rel(R.reshape(16), g["F"][i]) < 1e-13
        assert rel(J.reshape(16, 16), g["J"][i]) < 1e-13


def test_stokes_closed_form_matches_golden():
    g = golden("element_stokes.npz")
    A = el.stokes_element(g["X"]).reshape(-1, 16, 16)
    assert rel(A, g["A"]) < 1e-13


def test_literal_forms_regenerate_golden():
    """Golden vectors are reproducible from the committed generator code."""
    g = golden("element_ns.npz")
    for i in (0, 5):
        F, J = fl.ns_residual_and_jacobian_literal(g["X"][i], g["W"][i].reshape(16), float(g["Re"][i]))
        assert rel(F, g["F"][i]) < 1e-14 and rel(J, g["J"][i]) < 1e-14


@pytest.mark.parametrize("Re", [1.0, 100.0])
def test_jacobian_is_derivative_of_residual(Re):
    rng = np.random.default_rng(3)
    X = rng.normal(size=(5, 4, 3))
    W = rng.normal(size=(5, 4, 4))
    _, J = el.ns_element(X, W, Re)
    J = J.reshape(5, 16, 16)
    eps = 1e-6
    for k in range(16):
        dW = np.zeros((5, 16)); dW[:, k] = eps
        Rp, _ = el.ns_element(X, W + dW.reshape(5, 4, 4), Re, want_jac=False)
        Rm, _ = el.ns_element(X, W - dW.reshape(5, 4, 4), Re, want_jac=False)
        fd = ((Rp - Rm) / (2 * eps)).reshape(5, 16)
        assert np.abs(fd - J[:, :, k]).max() < 1e-6 * max(1.0, np.abs(J).max())


def test_metric_depends_on_local_vertex_zero():
    """G = K^T K changes with the choice of cell-local vertex 0, so
    the four rotations of one tet give different tau and different matrices."""
    g = golden("element_ns.npz")
    J0 = g["J"][0]
    diffs = []
    for rot in range(1, 4):
        p = np.roll(np.arange(4), -rot)
        perm = (4 * p[:, None] + np.arange(4)[None]).ravel()
        Jr = g["J"][rot]                         # tet with vertices rolled by rot
        # undo the dof relabelling: rolled local a <-> original p[a]
        back = np.empty_like(Jr)
        back[np.ix_(perm, perm)] = Jr
        diffs.append(rel(back, J0))
    assert max(diffs) > 1e-3                     # genuinely different operators
    # while the Stokes matrix (no G) is invariant under the relabelling
    s = golden("element_stokes.npz")
    p = np.roll(np.arange(4), -1)
    perm = (4 * p[:, None] + np.arange(4)[None]).ravel()
    back = np.empty((16, 16)); back[np.ix_(perm, perm)] = s["A"][1]
    assert rel(back, s["A"][0]) < 1e-13


def test_translation_invariance_and_stokes_rowsums():
    rng = np.random.default_rng(5)
    X = rng.normal(size=(3, 4, 3)); W = rng.normal(size=(3, 4, 4))
    R0, J0 = el.ns_element(X, W, 20.0)
    R1, J1 = el.ns_element(X + np.array([3.0, -2.0, 0.5]), W, 20.0)
    assert rel(R1, R0) < 1e-12 and rel(J1, J0) < 1e-11
    A = el.stokes_element(X)
    # velocity-velocity and pressure-pressure blocks annihilate constants (sum_b grad phi_b = 0)
    assert np.abs(A[:, :, 0, :, 0].sum(axis=2)).max() < 1e-12
    assert np.abs(A[:, :, 3, :, 3].sum(axis=2)).max() < 1e-12
    # coupling blocks are minus each other's transpose: A[(a,i),(b,p)] = -A[(b,p),(a,i)]
    assert np.abs(A[:, :, 0, :, 3] + np.transpose(A[:, :, 3, :, 0], (0, 2, 1))).max() < 1e-13


def test_reference_convection_slip_is_reproduced_not_fixed():
    """dot(u, grad(u)) != (u.grad)u; the oracle follows the reference."""
    g = golden("element_ns.npz")
    assert rel(g["F_corrected"], g["F"]) > 1e-4
    assert rel(g["J_corrected"], g["J"]) > 1e-4
