"""Round 4 prototype (CPU, scipy; test infrastructure): the damping estimate of the block-Jacobi smoother on the case the automatic
damping of round 3 fails on (jittered 120 x 30 x 30 duct, Re 200: tests/test_gpu_parity.py::test_damping_backoff_rescues_a_failed_linear_solve).
Restates the product's hierarchy (oracle/amg_cycle.py) and compares, per level, (a) the power-iteration estimate |lambda|max(S A) of
rounds 1-3 with (b) the Ritz values of a short Arnoldi process on S A and the damping limits they imply, then runs BiCGStab with each.

    python oracle/experiments/r4_damping.py 120 30 30 200 0.2 [block]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import amg_cycle as AC, cport  # noqa: E402
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M  # noqa: E402


def power(L, its=12):
    n = L.A.shape[0]
    i = np.arange(n)
    x = 1.0 + ((i * 2654435761) % 4294967296 >> 22) / 1024.0
    lam = []
    for _ in range(its):
        z = L.S @ (L.A @ x)
        lam.append(np.linalg.norm(z) / np.linalg.norm(x))
        x = z / np.linalg.norm(z)
    return max(lam[-3:]), x


def arnoldi(L, m=10, v0=None):
    n = L.A.shape[0]
    if v0 is None:
        i = np.arange(n)
        v0 = 1.0 + ((i * 2654435761) % 4294967296 >> 22) / 1024.0
    V = np.zeros((m + 1, n))
    H = np.zeros((m + 1, m))
    V[0] = v0 / np.linalg.norm(v0)
    for j in range(m):
        w = L.S @ (L.A @ V[j])
        for _ in range(2):
            h = V[:j + 1] @ w
            w = w - V[:j + 1].T @ h
            H[:j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(w)
        V[j + 1] = w / H[j + 1, j]
    return H[:m, :m]


def fov_limit(Hm, nang=64):
    """min over the boundary of the field of values W(H) of 2 Re z / |z|^2 (points with |z| >= 0.3 numerical radius)"""
    pts = []
    for t in np.linspace(0, 2 * np.pi, nang, endpoint=False):
        Ht = np.exp(1j * t) * Hm
        w, Vv = np.linalg.eigh((Ht + Ht.conj().T) / 2)
        x = Vv[:, -1]
        pts.append(x.conj() @ Hm @ x)
    pts = np.array(pts)
    r = np.abs(pts).max()
    sel = np.abs(pts) >= 0.3 * r
    return float(np.min(2 * pts[sel].real / np.abs(pts[sel]) ** 2)), r, pts


def solve(A, b, lv, sweeps, om, label, maxiter=400):
    its = [0]
    hist = []
    def cb(xk):
        its[0] += 1
    M_ = spla.LinearOperator(A.shape, matvec=lambda v: AC.cycle(lv, 0, v, sweeps, om))
    t0 = time.time()
    x, info = spla.bicgstab(A, b, rtol=1e-8, atol=0.0, M=M_, maxiter=maxiter, callback=cb)
    rel = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    print(f"{label:58s} its {its[0]:4d} info {info} rel {rel:.1e} omega {[round(o, 3) for o in om]} {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    cells = tuple(int(a) for a in sys.argv[1:4])
    Re = float(sys.argv[4])
    jit = float(sys.argv[5])
    block = len(sys.argv) > 6 and sys.argv[6] == "block"
    m = M.duct_mesh(cells, 4.0, jitter=jit)
    mask, g = B.duct_bcs(m).flatten()
    rp, ci = cport.pattern(m.num_nodes, m.tets)
    t0 = time.time()
    vals, F0 = cport.assemble("stokes", m.points, m.tets, None, Re, mask, g, rp, ci)
    U, its, reason, rn = cport.solve(m.num_nodes, rp, ci, vals, -F0, method="bicgstab", pc="ilu0", rtol=1e-10, maxit=3000)
    bm = mask.astype(bool)
    U[bm] = g[bm]
    print(f"stokes: {its} its reason {reason} {time.time() - t0:.0f}s", flush=True)
    vals, F = cport.assemble("ns", m.points, m.tets, U, Re, mask, g, rp, ci)
    A = cport.to_scipy(m.num_nodes, rp, ci, vals)
    b = -F
    nl_blk = (1, 2, 3, 4) if block else ()
    lv = AC.build(A, ~bm, dense_rows=512 if block else 0, block_levels=nl_blk)
    print("levels", [L.n for L in lv], f"{time.time() - t0:.0f}s", flush=True)
    nl = len(lv)
    if block:
        sweeps = [(1, 1), (1, 3), (2, 2), (1, 1), (1, 1), (1, 1)]
    else:
        sweeps = [(1, 1), (1, 6), (6, 6), (2, 2), (2, 2), (2, 2), (2, 2), (2, 2)]
    om_pow, om_ritz, om_fov = [], [], []
    RITZ = []
    for l, L in enumerate(lv[:-1]):
        lam, xdom = power(L)
        Hm = arnoldi(L, 10)
        th = np.linalg.eigvals(Hm)
        RITZ.append(th)
        big = th[np.abs(th) >= 0.5 * np.abs(th).max()]
        lim_ritz = float(np.min(2 * big.real / np.abs(big) ** 2))
        lim_fov, r, pts = fov_limit(Hm)
        print(f"level {l}: n {L.n}  power |lambda|max {lam:.3f} -> omega {min(0.8, 4 / (3 * lam)):.3f} | Ritz max |theta| "
              f"{np.abs(th).max():.3f} (max imag {np.abs(th.imag).max():.3f}) stability limit {lim_ritz:.3f} -> 2/3 of it "
              f"{min(0.8, 2 * lim_ritz / 3):.3f} | field of values: radius {r:.3f}, limit {lim_fov:.3f} -> 2/3 of it {min(0.8, 2 * lim_fov / 3):.3f}",
              flush=True)
        om_pow.append(min(0.8, 4 / (3 * lam)))
        om_ritz.append(min(0.8, 2 * lim_ritz / 3))
        om_fov.append(min(0.8, 2 * lim_fov / 3))
    om_pow.append(0.8); om_ritz.append(0.8); om_fov.append(0.8)
    if not os.environ.get("R4_SKIP_BASE"):
        solve(A, b, lv, sweeps, om_pow, "power iteration (rounds 1-3)")
        solve(A, b, lv, sweeps, om_ritz, "Arnoldi(10) Ritz values, 2/3 of the stability limit")
        solve(A, b, lv, sweeps, om_fov, "field of values of H_10, 2/3 of its limit")
        solve(A, b, lv, sweeps, [0.7 * o for o in om_pow], "power iteration x 0.7 (the retry)")
    # --- scan: the largest damping whose amplification of every Ritz value stays below g (capped by the power-iteration rule)
    def rule(g):
        om = []
        for l, L in enumerate(lv[:-1]):
            th = RITZ[l]
            big = th[np.abs(th) >= 0.3 * np.abs(th).max()]
            lim = np.min((big.real + np.sqrt(big.real ** 2 + (g * g - 1.0) * np.abs(big) ** 2)) / np.abs(big) ** 2)
            om.append(float(min(om_pow[l], lim)))
        return om + [0.8]
    for g in (1.1, 1.2, 1.3):
        solve(A, b, lv, sweeps, rule(g), f"Ritz rule: amplification of any Ritz value <= {g}")
    # --- one level at a time around the set that works (power iteration x 0.7)
    if os.environ.get("R4_SCAN"):
        base = [0.7 * o for o in om_pow]
        for l, vals_ in ((0, (0.235, 0.459)), (1, (0.30, 0.555, 0.68)), (2, (0.743,)), (3, (0.743,))):
            for v in vals_:
                om = list(base)
                om[l] = v
                solve(A, b, lv, sweeps, om, f"x 0.7 set with omega[{l}] = {v}")
        om = list(om_pow)
        om[1] = 0.476
        solve(A, b, lv, sweeps, om, "power-iteration set with omega[1] = 0.476 only")
        om = list(om_pow)
        om[1] = 0.476; om[2] = 0.52
        solve(A, b, lv, sweeps, om, "power-iteration set with omega[1] = 0.476, omega[2] = 0.52")
