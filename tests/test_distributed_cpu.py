"""N>1 path on CPU: world_size-2/3 gloo runs of the partition + halo plan that the
C-ABI consumes (sns_attach_comm), with the oracle doing the per-rank arithmetic.
Checks: redundant ghost-tet assembly reproduces the owned rows exactly (no assembly
communication needed), halo exchange + local SpMV == global SpMV, all-reduced dots,
and a distributed block-Jacobi BiCGStab reaches the serial solution."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import assemble as asm, solve as S
        if kind == "duct":
            m = M.duct_mesh((8, 3, 3), 4.0, jitter=0.1)
            mask, g = B.duct_bcs(m).flatten()
        else:
            m = M.cavity_mesh(5)
            mask, g = B.cavity_bcs(m).flatten()
        owner = PT.rcb_partition(m.points, world)
        part = PT.build_local_part(m, mask, g, owner, rank, world)
        rng = np.random.default_rng(42)
        w = rng.normal(size=m.num_dofs) * 0.3
        Re = 15.0
        Jg, Fg = asm.assemble_ns(m.points, m.tets, w, Re, mask, g)
        wl = PT.scatter_global(part, w)
        Jl, Fl = asm.assemble_ns(part.mesh.points, part.mesh.tets, wl, Re, part.bc_mask, part.bc_val)
        no = 4 * part.n_owned
        gdof = (4 * part.l2g[:, None] + np.arange(4)[None]).ravel()
        # owned rows of the redundantly assembled local operator == global rows
        err_rows = abs(Jl[:no] - Jg[gdof[:no]][:, gdof]).max()
        err_F = np.abs(Fl[:no] - Fg[gdof[:no]]).max()
        # halo exchange + local SpMV
        x = rng.normal(size=m.num_dofs)
        xl = np.zeros(4 * part.n_local)
        xl[:no] = x[gdof[:no]]
        xt = torch.from_numpy(xl)
        PT.halo_exchange_torch(part, xt)
        err_halo = np.abs(xt.numpy() - x[gdof]).max()
        yl = Jl[:no] @ xt.numpy()
        err_spmv = np.abs(yl - (Jg @ x)[gdof[:no]]).max()
        # interior / boundary split of the multi-GPU SpMV (exchange_and_spmv in csrc/sns_api.hip): rows without a
        # ghost column are computed from the OWNED part of x alone, i.e. before the halo has arrived; only the rows
        # sns_host_boundary_rows lists wait for it
        from stabilized_navier_stokes_flow_fenicsx_amd import _lib
        rp, ci, _, _ = _lib.host_pattern(part.n_local, part.mesh.tets)
        bnd = _lib.host_boundary_rows(part.n_owned, rp, ci)
        interior = np.setdiff1d(np.arange(part.n_owned), bnd)
        x_owned_only = xl.copy()                                    # ghost tail still zero: no exchange yet
        di = (4 * interior[:, None] + np.arange(4)[None]).ravel()
        db = (4 * bnd[:, None] + np.arange(4)[None]).ravel()
        y_split = np.zeros(no)
        y_split[di] = Jl[di] @ x_owned_only
        y_split[db] = Jl[db] @ xt.numpy()                           # after the exchange
        err_split = np.abs(y_split - (Jg @ x)[gdof[:no]]).max()
        n_bnd = len(bnd)
        d = torch.tensor([float(xl[:no] @ xl[:no])], dtype=torch.float64)
        dist.all_reduce(d)
        err_dot = abs(float(d) - float(x @ x)) / float(x @ x)
        # distributed block-Jacobi BiCGStab on the Stokes system
        Ag, bg = asm.assemble_stokes(m.points, m.tets, mask, g)
        Al, bl_full = asm.assemble_stokes(part.mesh.points, part.mesh.tets, part.bc_mask, part.bc_val)
        Al = Al[:no]
        bl = bg[gdof[:no]]
        Dinv = S.block_jacobi_inverse(Ag)[part.l2g[: part.n_owned]]

        def Mv(v):
            return np.einsum("nij,nj->ni", Dinv, v.reshape(-1, 4)).ravel()

        def Av(v):
            t = torch.zeros(4 * part.n_local, dtype=torch.float64)
            t[:no] = torch.from_numpy(v)
            PT.halo_exchange_torch(part, t)
            return Al @ t.numpy()

        def gdot(a, b):
            t = torch.tensor([float(a @ b)], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t)

        def gdots(*pairs):                                          # ONE all-reduce for several dots
            t = torch.tensor([float(a @ b) for a, b in pairs], dtype=torch.float64)
            dist.all_reduce(t)
            n_allreduce[0] += 1
            return t.numpy()

        # the product's latency-lean BiCGStab (csrc/sns_api.hip:bicgstab): TWO all-reduces per iteration --
        # <rhat, v>, then (t.s, t.t, rhat.s, rhat.t, s.s) from which omega, the next rho and ||r||^2 follow
        n_allreduce = [0]
        xk = np.zeros(no)
        r = bl - Av(xk)
        rhat = r.copy()
        rho, bb = gdots((rhat, r), (bl, bl))
        bn = np.sqrt(bb)
        n_allreduce[0] = 0
        beta = 0.0
        omega = 1.0
        v = np.zeros(no)
        p = np.zeros(no)
        its = 0
        for its in range(1, 400):
            p = r + beta * (p - omega * v)
            ph = Mv(p)
            v = Av(ph)
            alpha = rho / gdots((rhat, v))[0]
            s = r - alpha * v
            sh = Mv(s)
            t = Av(sh)
            ts, tt, hs, ht, ss = gdots((t, s), (t, t), (rhat, s), (rhat, t), (s, s))
            omega = ts / tt
            rho_new = hs - omega * ht
            rr = ss - 2 * omega * ts + omega * omega * tt
            xk = xk + alpha * ph + omega * sh
            r = s - omega * t
            beta = (rho_new / rho) * (alpha / omega)
            rho = rho_new
            if np.sqrt(max(rr, 0.0)) <= 1e-10 * bn:
                break
        allreduce_per_it = n_allreduce[0] / its
        err_rr = abs(np.sqrt(max(rr, 0.0)) - np.sqrt(gdot(r, r))) / bn
        Ug = S.lu_solve(Ag, bg)
        err_sol = np.abs(xk - Ug[gdof[:no]]).max() / np.abs(Ug).max()
        xo, its_serial, _ = S.bicgstab_bj(Ag, bg, rtol=1e-10)
        # gather_owned round trip
        full = PT.gather_owned(part, torch.from_numpy(PT.scatter_global(part, w)), m.num_nodes)
        err_gather = np.abs(full.numpy() - w).max()
        q.put((rank, dict(err_rows=err_rows, err_F=err_F, err_halo=err_halo, err_spmv=err_spmv, err_dot=err_dot,
                          err_sol=err_sol, its=its, its_serial=its_serial, err_gather=err_gather,
                          err_split=err_split, n_bnd=n_bnd, allreduce_per_it=allreduce_per_it, err_rr=err_rr,
                          n_owned=part.n_owned, n_local=part.n_local, nbr=list(map(int, part.neighbors)))))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, dict(error=traceback.format_exc())))


@pytest.mark.parametrize("world,kind", [(2, "duct"), (3, "cavity")])
def test_partition_halo_and_distributed_krylov_gloo(world, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    tot_owned = 0
    for r in range(world):
        out = res[r]
        assert "error" not in out, out.get("error")
        assert out["err_rows"] < 1e-13 and out["err_F"] < 1e-13
        assert out["err_halo"] == 0.0 and out["err_spmv"] < 1e-12 and out["err_dot"] < 1e-13
        assert out["err_sol"] < 1e-6 and abs(out["its"] - out["its_serial"]) <= 3
        assert out["err_gather"] == 0.0
        assert out["err_split"] < 1e-12 and 0 < out["n_bnd"] < out["n_owned"]    # interior rows need no halo
        assert out["allreduce_per_it"] == 2.0 and out["err_rr"] < 1e-12          # merged reductions, same residual
        assert out["n_local"] > out["n_owned"] and len(out["nbr"]) >= 1
        tot_owned += out["n_owned"]
    n_nodes = (9 * 4 * 4) if kind == "duct" else 6 ** 3
    assert tot_owned == n_nodes                                   # every node owned exactly once


def test_rcb_gives_x_slabs_on_the_duct():
    m = M.duct_mesh((16, 3, 3), 4.0)
    owner = PT.rcb_partition(m.points, 4)
    assert np.bincount(owner).tolist() == [68, 68, 68, 68]
    xmax = [m.points[owner == r, 0].max() for r in range(4)]
    xmin = [m.points[owner == r, 0].min() for r in range(4)]
    assert all(xmax[r] <= xmin[r + 1] + 1e-12 for r in range(3))   # slabs ordered in x
    mask, g = B.duct_bcs(m).flatten()
    parts = [PT.build_local_part(m, mask, g, owner, r, 4) for r in range(4)]
    assert [len(p.neighbors) for p in parts] == [1, 2, 2, 1]       # <= 2 neighbours per GPU
    for p in parts:                                                # send/recv plans are mutually consistent
        for k, r in enumerate(p.neighbors):
            q = parts[r]
            kk = list(q.neighbors).index(p.rank)
            sent = p.l2g[p.send_idx[p.send_ptr[k]:p.send_ptr[k + 1]]]
            recv = q.l2g[q.recv_idx[q.recv_ptr[kk]:q.recv_ptr[kk + 1]]]
            assert np.array_equal(sent, recv)


@pytest.mark.parametrize("cells,length,nranks", [((12, 3, 4), 4.0, 2), ((13, 3, 2), 4.0, 3), ((16, 2, 3), 8.0, 4),
                                                 ((12, 5, 5), 1.0, 4)])
def test_slab_part_without_global_mesh_matches_global_partition(cells, length, nranks):
    """bench.py's weak-scaling runs mesh only the rank's own slab (partition.duct_slab_part); the result must be
    the LocalPart the global mesh + partitioner would give: ids, ordering, halo plans, coordinates, BC data."""
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT
    m = M.duct_mesh(cells, length)
    mask, g = B.duct_bcs(m).flatten()
    own = PT.slab_owner(m.num_nodes, nranks)
    if length / nranks >= 1.0:                       # long slabs: RCB degenerates to the same x-slabs
        assert np.array_equal(own, PT.rcb_partition(m.points, nranks))
    for r in range(nranks):
        a = PT.build_local_part(m, mask, g, own, r, nranks)
        c = PT.duct_slab_part(cells, length, r, nranks)
        assert a.n_owned == c.n_owned
        for k in ("l2g", "tet_ids", "bc_mask", "bc_val", "neighbors", "send_ptr", "send_idx", "recv_ptr", "recv_idx"):
            assert np.array_equal(getattr(a, k), getattr(c, k)), k
        assert np.array_equal(a.mesh.tets, c.mesh.tets) and np.array_equal(a.mesh.points, c.mesh.points)


def _run_bench(args, env_extra=None, timeout=240):
    """bench.py as a plain command (what the driver types), from the repo root; returns (rc, stdout, stderr)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, capture_output=True,
                       text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` as a plain command starts TWO ranks itself (torch.distributed.run on 127.0.0.1) --
    here in --dry-run mode: rendezvous, the per-rank slab partition, the halo plans checked by a real exchange of
    global node ids, the boundary-row split, one all-reduce -- all on gloo, no HIP call.  The line says n_gpus 2."""
    import json
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6"])
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None
    c = d["config"]
    assert c["halo_plans_consistent_ranks"] == 2                       # both ranks received exactly their ghosts' ids
    assert c["owned_nodes_total"] == 25 * 7 * 7                        # every node owned once
    assert 2 * 49 <= c["boundary_rows_total"] < 4 * 49 and c["neighbours_of_rank0"] == [1]
    # an RCB-partitioned config through the same path
    rc, out, err = _run_bench(["--gpus", "3", "--dry-run", "--config", "3", "--cells", "6,6,6"])
    assert rc == 0, err[-2000:]
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["config"]["halo_plans_consistent_ranks"] == 3 and d["config"]["owned_nodes_total"] == 343


def test_bench_repeats_a_crashed_launch_once_without_the_two_stream_overlap(tmp_path):
    """A first attempt that fails fast is repeated ONCE with SNS_NO_OVERLAP=1 (halo exchange and operator pass on one
    stream) and the result line says so; a launch that works carries launch_fallback = null."""
    import json
    # (the first attempt's record goes to a directory of the test's own: a CPU test must not leave a file in the repo's
    # gpurun_out/ that reads like a GPU-side crash record -- VERDICT r4)
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6"],
                              {"SNS_DRYRUN_CRASH_WITH_OVERLAP": "1", "SNS_BENCH_LOG_DIR": str(tmp_path)})
    assert rc == 0, err[-2000:]
    assert "one more attempt with SNS_NO_OVERLAP=1" in err
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "SNS_NO_OVERLAP=1 after a first attempt that exited" in d["launch_fallback"]
    # the fallback is not a silent success -- the line is marked degraded and the first attempt's exit code and
    # stderr tail are on file
    assert d["degraded"] is True
    rec = d["launch_fallback"].split("record: ")[1].rstrip(")")
    assert os.path.dirname(rec) == str(tmp_path)
    assert os.path.exists(rec) and "first attempt (overlapped halo) exited" in open(rec).read()
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6"])
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert rc == 0 and d["launch_fallback"] is None and d["degraded"] is False


def test_bench_weak_leg_deadline_keeps_the_headline_line():
    """a weak-scaling leg that hangs (here: on both ranks, for good) must not cost the headline.  After
    --weak-timeout seconds rank 0 prints THE line with weak_scaling = {error: timeout ...}, every rank leaves with exit code 0
    (no re-exec, no child), and the launcher reports success with exactly one line."""
    import json
    import time
    t0 = time.time()
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6", "--weak-timeout", "3"],
                              {"SNS_DRYRUN_WEAK_STALL": "1", "SNS_WATCHDOG_S": "600"}, timeout=300)
    assert rc == 0, err[-2000:]
    assert time.time() - t0 < 200
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "timeout" in d["weak_scaling"]["error"]
    assert "weak-scaling leg exceeded" in err
    # in-process: a leg that finishes in time returns normally and leaves the line to the caller
    import bench
    o = {"weak_scaling": None}
    bench.run_weak_leg_guarded(o, 0, 30.0, lambda: o.__setitem__("weak_scaling", {"value": 1.0}))
    assert o["weak_scaling"] == {"value": 1.0}


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert rc == 2 and "WORLD_SIZE" in err and out.strip() == ""


def test_bench_secondary_legs_share_one_budget():
    """headline + peer leg + weak leg must fit the driver's clock (600 s) whatever the legs do.  The legs take
    their deadlines from ONE --budget (an equal share of what is left, at most their own cap; too little left = skipped), so the
    one line is out and every rank has exited 0 inside it.  Here with real ranks (dry run, both legs never come back): the
    peer leg's share of a 24-s budget expires, rank 0 prints THE line, both ranks exit 0 -- well inside the budget; and with a
    budget that leaves no room the legs are skipped and say so."""
    import json
    import time
    env = {"SNS_DRYRUN_PEER_STALL": "1", "SNS_DRYRUN_WEAK_STALL": "1", "SNS_WATCHDOG_S": "600", "SNS_BENCH_MIN_LEG_S": "2",
           "SNS_BENCH_RESERVE_S": "4"}
    t0 = time.time()
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6", "--budget", "24", "--peer-timeout", "300",
                               "--weak-timeout", "300"], env, timeout=200)
    took = time.time() - t0
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert "timeout" in d["peer_transport"]["error"] and "peer-transport leg exceeded" in err
    assert took < 24 + 25, took                     # (the budget + the launcher's own start-up and teardown)
    # no room at all: both legs are skipped, the line still comes
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6", "--budget", "1"], env, timeout=200)
    assert rc == 0, err[-2000:]
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert "skipped" in d["peer_transport"] and "skipped" in d["weak_scaling"]
    # the arithmetic itself: equal shares of what is left, capped, never below the minimum
    import bench
    now = bench.T_START + 100.0
    assert bench.leg_seconds(540.0, 300.0, 2, now=now) == (540.0 - 100.0 - bench.LEG_RESERVE_S) / 2
    assert bench.leg_seconds(540.0, 150.0, 2, now=now) == 150.0
    assert bench.leg_seconds(540.0, 300.0, 1, now=bench.T_START + 540.0 - bench.LEG_RESERVE_S - 5.0) == 0.0
    # headline 200 s + peer leg at its share + weak leg at its share stay inside 540 s
    t_peer = bench.leg_seconds(540.0, 300.0, 2, now=bench.T_START + 200.0)
    t_weak = bench.leg_seconds(540.0, 300.0, 1, now=bench.T_START + 200.0 + t_peer)
    assert 200.0 + t_peer + t_weak <= 540.0 - bench.LEG_RESERVE_S + 1e-9


def test_bench_shared_gpu_rehearsal_needs_the_peer_transport():
    """--shared-gpu puts every rank on cuda:0, which RCCL cannot do: without --transport peer the bench refuses before it touches a GPU"""
    rc, out, err = _run_bench(["--shared-gpu"])
    assert rc == 2 and "--transport peer" in err and out.strip() == ""


def test_bench_guarded_leg_reports_under_its_own_key():
    """the peer-transport leg runs under the weak leg's guard with its own key: what the leg stores is what the line carries"""
    import bench
    o = {"peer_transport": None, "weak_scaling": None}
    bench.run_weak_leg_guarded(o, 0, 30.0, lambda: o.__setitem__("peer_transport", {"value": 2.0}), key="peer_transport",
                               what="peer-transport")
    assert o == {"peer_transport": {"value": 2.0}, "weak_scaling": None}


def test_bench_watchdog_turns_a_stalled_rank_into_a_nonzero_exit():
    """A rank that stops making progress exits non-zero by itself (no re-exec); the launcher passes the failure on."""
    import time
    t0 = time.time()
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--cells", "24,6,6"],
                              {"SNS_DRYRUN_STALL_RANK": "1", "SNS_WATCHDOG_S": "2"}, timeout=120)
    assert rc != 0 and "watchdog: rank 1" in err
    assert time.time() - t0 < 100
