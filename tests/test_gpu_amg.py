"""Round 4: the pieces of the AMG preconditioner that are new this round, each against a CPU restatement under oracle/:
  * the blocked Gauss-Jordan dense inverse on the fp64 matrix cores (csrc/sns_dense.hip)      vs numpy.linalg.inv
  * the V-cycle with the dense coarsest level and the aggregate-block Jacobi smoother (csrc/sns_block.hip)
                                                                                              vs oracle/amg_cycle.py (scipy)
  * the stagnation watch of the damping retry
Tolerances: the cycle's matrix data are fp32 copies here (amg_f32_matrix = 1: level matrices, D^-1 / B^-1, the dense inverse;
vectors and arithmetic fp64), the oracle is all fp64: 1e-5 relative on a random vector (observed ~1e-7); with the default fp16
copies the same comparison is a bound on the perturbation (< 0.1, as in test_fused_post_sweep_is_the_same_preconditioner).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
    return FlowProblem


@pytest.mark.parametrize("N", [1, 5, 63, 64, 65, 200, 1000, 1900])
def test_blocked_gauss_jordan_inverse_vs_numpy(N):
    """sns_dense_inverse (64 x 64 blocks, v_mfma_f64_16x16x4_f64 rank-64 updates, no pivoting) on matrices of the class it is
    meant for -- positive definite symmetric part, skew part several times larger (convection-dominated coarse operators) --
    against numpy's LAPACK inverse: |X A - I| and the difference to numpy at fp64 round-off times the condition number."""
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(N)
    S = rng.normal(size=(N, N))
    A = 3.0 * (S - S.T) + np.diag(1.0 + rng.random(N)) * np.sqrt(N) + 0.1 * (S + S.T)
    A[N // 2, :] = 0.0
    A[:, N // 2] = 0.0
    A[N // 2, N // 2] = 1.0                                    # an identity row + column (a Dirichlet dof)
    Ad = torch.from_numpy(A).cuda()
    Xd = torch.empty_like(Ad)
    assert lib.sns_dense_inverse(0, N, Ad.data_ptr(), Xd.data_ptr()) == 0
    X = Xd.cpu().numpy()
    ref = np.linalg.inv(A)
    cond = np.linalg.cond(A)
    assert np.abs(X @ A - np.eye(N)).max() < 1e-13 * cond * N
    assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-13 * cond
    # a singular matrix is reported, not inverted
    Z = torch.zeros((max(N, 2), max(N, 2)), dtype=torch.float64, device="cuda")
    assert lib.sns_dense_inverse(0, Z.shape[0], Z.data_ptr(), torch.empty_like(Z).data_ptr()) != 0


def _problem(gpu, opts):
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
    m = M.duct_mesh((40, 10, 10), 4.0, jitter=0.15)
    mask, g = B.duct_bcs(m).flatten()
    P = gpu(m, (mask, g), reynolds=60.0, **opts)
    U, r = P.stokes_solve()
    assert r.reason > 0
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    return m, mask, P, U, F


def _oracle_cycle(P, mask, r, block, dense_rows):
    from oracle import amg_cycle as AC
    import scipy.sparse as sp
    A = P.to_scipy()
    rp, ci, _ = P.bsr()                                          # the pattern the product aggregates on (explicit zero blocks too)
    rp, ci = rp.cpu().numpy(), ci.cpu().numpy()
    G = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(len(rp) - 1, len(rp) - 1))
    hier = P.hierarchy()
    cyc = P.cycle()                                              # smoother kind and sweeps per level, as the GPU runs them
    nl = len(hier)
    blk = tuple(l for l, c in enumerate(cyc) if c["kind"] == 1)
    assert blk == (tuple(range(1, nl - 1)) if block else ()), cyc
    assert cyc[-1]["kind"] == (3 if dense_rows else 2), cyc
    lv = AC.build(A, ~mask.astype(bool), dense_rows=dense_rows, block_levels=blk, graph=G)
    assert [L.n for L in lv] == [h["rows"] for h in hier], ([L.n for L in lv], hier)
    sweeps = [(c["pre"], c["post"]) for c in cyc]
    om = [h["omega"] for h in hier]
    return AC.cycle(lv, 0, r, sweeps, om), lv


@pytest.mark.parametrize("block,dense_rows", [(0, 0), (0, 100), (1, 0), (1, 100)])
def test_vcycle_matches_the_scipy_restatement(gpu, block, dense_rows):
    """pc_apply of the HIP V-cycle == oracle/amg_cycle.py's restatement of the same cycle (same aggregates, Galerkin operators,
    sweeps, the GPU's own damping), for nodal-block and aggregate-block smoothing and for the coarsest level solved by the
    one-workgroup inverse (dense_rows 0: 10 nodes) resp. the blocked Gauss-Jordan inverse (dense_rows 100: 78 nodes)."""
    m, mask, P, U, F = _problem(gpu, dict(amg_block_smooth=block, amg_dense_rows=dense_rows, amg_f32_matrix=1))
    P.pc_setup()
    hier = P.hierarchy()
    assert len(hier) == (4 if dense_rows == 0 else 3), hier
    r = np.random.default_rng(5).normal(size=m.num_dofs)
    z = P.pc_apply(torch.from_numpy(r).cuda()).cpu().numpy()
    zo, lv = _oracle_cycle(P, mask, r, block, dense_rows)
    d32 = rel(z, zo)
    # and the cycle is a fixed linear operator
    z2 = P.pc_apply(torch.from_numpy(-2.0 * r).cuda()).cpu().numpy()
    assert rel(z2, -2.0 * z) < 1e-12
    # the default low-precision format: a bounded perturbation of the same operator, same Krylov behaviour
    y32, k32 = P.krylov_solve(F)
    P.set_options(amg_f32_matrix=2)
    P.pc_setup()
    z16 = P.pc_apply(torch.from_numpy(r).cuda()).cpu().numpy()
    y16, k16 = P.krylov_solve(F)
    print(f"  block {block} dense_rows {dense_rows}: levels {[h['rows'] for h in hier]} sweeps {[h['sweeps'] for h in hier]} "
          f"HIP vs scipy cycle fp32 copies {d32:.2e}, fp16 copies {rel(z16, zo):.2e}; its {k32.its} / {k16.its}")
    assert d32 < 1e-5
    assert rel(z16, zo) < 0.1
    assert k32.reason > 0 and k16.reason > 0 and abs(k32.its - k16.its) <= 2
    assert rel(y16.cpu().numpy(), y32.cpu().numpy()) < 1e-6
    P.close()


def test_block_smoother_and_dense_level_solve_to_the_same_fields(gpu):
    """The round-4 cycle (aggregate blocks + dense coarsest level, the defaults) and the round-3 cycle (nodal blocks, hierarchy down
    to <= 32 nodes) are two preconditioners of the same system: Newton converges to the same fields (< 1e-6, north_star's bound
    against the reference), in the same number of Newton iterations, and the new cycle does not need more Krylov iterations."""
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
    m = M.duct_mesh((64, 16, 16), 4.0, jitter=0.1)
    bcs = B.duct_bcs(m)
    out = {}
    for name, opts in (("r4", {}), ("r3", dict(amg_block_smooth=0, amg_dense_rows=0))):
        P = gpu(m, bcs, reynolds=100.0, **opts)
        U, r = P.stokes_solve()
        w, n = P.newton_solve(U.clone())
        assert r.reason > 0 and n.reason > 0
        out[name] = (w.cpu().numpy(), n.its, n.ksp_its, [h["rows"] for h in P.hierarchy()], [h["sweeps"] for h in P.hierarchy()])
        P.close()
    print(f"  r4 levels {out['r4'][3]} sweeps {out['r4'][4]} ksp its {out['r4'][2]}; r3 levels {out['r3'][3]} sweeps {out['r3'][4]} "
          f"ksp its {out['r3'][2]}")
    assert rel(out["r4"][0], out["r3"][0]) < 1e-6
    assert out["r4"][1] == out["r3"][1]
    assert out["r4"][2] <= 1.15 * out["r3"][2] + 2
    assert len(out["r4"][3]) < len(out["r3"][3])


def test_slow_but_converging_solve_is_not_retried(gpu):
    """the stagnation watch of amg_retry_damping must not end a solve that is converging slowly.  With a
    deliberately weak smoother (damping 0.15) BiCGStab needs several times the usual iterations but keeps setting new best
    residuals; a watch window of 12 iterations -- under round 3's rule (best residual must HALVE within the window) this solve
    was ended as a 'breakdown' and restarted with the damping scaled by 0.7 -- leaves it alone: no retry, converged, the
    damping factor untouched.  A solve that really stagnates is still caught (test_damping_backoff_rescues_a_failed_linear_solve)."""
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
    m = M.duct_mesh((48, 12, 12), 4.0, jitter=0.1)
    P = gpu(m, B.duct_bcs(m), reynolds=50.0, amg_omega=0.15, amg_retry_stall_its=12)
    U, r = P.stokes_solve()
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    P.reset_timings()
    y, k = P.krylov_solve(F)
    c = P.counters()
    P.set_options(amg_omega=0.8)
    y2, k2 = P.krylov_solve(F)
    print(f"  weak smoother: {k.its} its (reason {k.reason}), default damping: {k2.its} its; retries {c['damping_retries']}")
    assert k.reason > 0 and k.its > 1.5 * k2.its
    assert c["damping_retries"] == 0 and c["damping_factor"] == 1.0 and c["first_attempt_reason"] == 0
    P.close()


@pytest.mark.parametrize("block,dense_rows,fmt", [(1, 100, 2), (1, 0, 1), (0, 0, 2)])
def test_fused_residual_restriction_is_the_same_cycle(gpu, block, dense_rows, fmt):
    """amg_fuse_restrict: below the fine level residual, restriction and the next level's first sweep are one launch
    (k_resid_restrict) instead of k_spmv_lp + k_restrict / k_restrict_blk.  Same row products, the member sums in the same order:
    the cycle is the same linear operator to round-off (observed: bitwise), for nodal and aggregate-block coarse levels, a dense or a
    smoothed last level, fp32 and fp16 matrix copies -- and the Krylov solve takes the same iterations."""
    m, mask, P, U, F = _problem(gpu, dict(amg_block_smooth=block, amg_dense_rows=dense_rows, amg_f32_matrix=fmt, amg_fuse_restrict=1))
    P.pc_setup()
    r = torch.from_numpy(np.random.default_rng(11).normal(size=m.num_dofs)).cuda()
    z1 = P.pc_apply(r).cpu().numpy()
    y1, k1 = P.krylov_solve(F)
    P.set_options(amg_fuse_restrict=0)
    z0 = P.pc_apply(r).cpu().numpy()
    y0, k0 = P.krylov_solve(F)
    print(f"  block {block} dense_rows {dense_rows} fmt {fmt}: levels {[h['rows'] for h in P.hierarchy()]} fused vs separate {rel(z1, z0):.1e}, "
          f"its {k1.its} / {k0.its}")
    assert rel(z1, z0) < 1e-13
    assert k1.reason > 0 and k0.reason > 0 and k1.its == k0.its
    P.close()
