"""Round 4: the peer-window transport (sns_peer_*, csrc/sns_comm.hip) between real processes.

RCCL cannot put two ranks on one GPU, so until now the partitioned solver had only run with ranks = threads of one process (the
Team emulation).  The peer transport needs nothing but HIP IPC: here every rank is a PROCESS with its own HIP context, the ranks
map each other's windows, and halo exchange / all-reduce / all-gather are the production kernels (stores into the peer's window,
sequence flags, bounded waits).  All ranks share the box's one GPU, so the xGMI hop itself is the only thing not exercised.

What is checked, per rank, against the serial solve of the same problem (the Team tests' bounds): SpMV with halo exchange 1e-12,
Stokes 1e-6, Newton fields 1e-8, identical iteration counts and decisions on every rank; and that a rank which never joins a
collective turns into SNS_E_COMM on the others after SNS_PEER_TIMEOUT_MS instead of a hang.
"""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(kind, world, tmp_path, env_extra=None, deadline=420.0):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    port = _free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    procs, outs, logs = [], [], []
    for r in range(world):
        o = str(tmp_path / f"rank{r}.json")
        lg = open(str(tmp_path / f"rank{r}.log"), "w")
        outs.append(o)
        logs.append(lg)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "peer_worker.py"), str(r), str(world), str(port), kind, o],
                                      stdout=lg, stderr=subprocess.STDOUT, env=env))
    t0 = time.time()
    try:
        for p in procs:
            p.wait(timeout=max(1.0, deadline - (time.time() - t0)))
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()                                      # exactly the processes started above
        for p in procs:
            p.wait()
        tails = [open(str(tmp_path / f"rank{r}.log")).read()[-1500:] for r in range(world)]
        pytest.fail(f"peer ranks did not finish within {deadline:.0f} s\n" + "\n---\n".join(tails))
    finally:
        for lg in logs:
            lg.close()
    res = []
    for r in range(world):
        if not os.path.exists(outs[r]):
            pytest.fail(f"rank {r} left no result (exit {procs[r].returncode}):\n" + open(str(tmp_path / f"rank{r}.log")).read()[-3000:])
        res.append(json.load(open(outs[r])))
    return res


@pytest.mark.parametrize("kind,world,opts", [("duct", 3, None), ("duct-rep-dense", 3, None), ("cavity", 4, None),
                                             # five ranks: an RCB partition that is not a power of two (with this process the six
                                             # GPU processes a box admits)
                                             ("cavity", 5, None),
                                             # round 4's form of the exchange (put + wait / unpack into the ghost tail, interior rows on
                                             # a second stream meanwhile): what RCCL-shaped code paths and halo_windows = 0 run
                                             ("cavity", 4, {"halo_windows": 0, "amg_exact_sweeps": 0})])
def test_partitioned_solve_between_processes_over_peer_windows(kind, world, opts, tmp_path):
    res = _run_ranks(kind, world, tmp_path, env_extra={"SNS_TEST_OPTS": json.dumps(opts)} if opts else None)
    for r in res:
        assert r["ok"], r
    print("  " + kind + ": " + "; ".join(f"rank {r['rank']}: {r['n_owned']} nodes, nbrs {r['neighbors']}, stokes {r['stokes_its']} its, "
                                       f"ksp {r['ksp_its']}, {r['exchanges']} exchanges, {r['allreduces']} all-reduces, {r['seconds']:.1f} s"
                                       for r in res))
    r0 = res[0]
    for r in res:
        assert r["transport"] == "peer" and r["nranks"] == world
        assert r["err_spmv"] < 1e-12 and r["err_stokes"] < 1e-6 and r["err_newton"] < 1e-8, r
        assert r["stokes_reason"] > 0 and r["newton_reason"] == r["serial"]["newton_reason"] and r["newton_its"] == r["serial"]["newton_its"]
        assert (r["stokes_its"], r["ksp_its"], r["levels"]) == (r0["stokes_its"], r0["ksp_its"], r0["levels"])   # same decisions everywhere
        assert r["stokes_its"] <= 2 * r["serial"]["stokes_its"] + 4
    # a partitioned handle with few fine rows per rank smooths its FINE level with the aggregate blocks too (amg_block_fine_rows)
    assert all(r["cycle"][0][0] == 1 for r in res), r0["cycle"]
    if kind == "duct-rep-dense":
        assert r0["levels"] == 2                               # fine level + the replicated, directly solved level 1
    if kind == "cavity":
        most = max(len(r["neighbors"]) for r in res)
        assert most == 3 if world == 4 else most >= 3


def test_carried_puts_match_separate_put_launches_between_processes(tmp_path):
    """Round 5: the put half of a halo exchange rides in the kernel that produces the vector (PutDst: stores into the neighbours'
    windows next to the local store, the last storing workgroup fences and raises the flags).  SNS_NO_CARRIED_PUT keeps the
    separate k_halo_put launch.  Four processes over mapped windows, both ways: the same iterations, decisions and errors."""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    ra = _run_ranks("cavity", 4, tmp_path / "a")
    rb = _run_ranks("cavity", 4, tmp_path / "b", env_extra={"SNS_NO_CARRIED_PUT": "1"})
    for a, b in zip(ra, rb):
        assert a["ok"] and b["ok"]
        for k in ("stokes_its", "ksp_its", "newton_its", "levels", "cycle", "err_spmv", "err_stokes", "err_newton"):
            assert a[k] == b[k], (k, a[k], b[k])
    # with the puts carried an exchange is counted all the same, only the launch is gone
    assert ra[0]["exchanges"] == rb[0]["exchanges"]


def test_a_rank_that_never_arrives_is_an_error_not_a_hang(tmp_path):
    res = _run_ranks("duct-late", 3, tmp_path, env_extra={"SNS_PEER_TIMEOUT_MS": "3000"}, deadline=300.0)
    late = [r for r in res if r.get("skipped")]
    waited = [r for r in res if not r.get("skipped")]
    assert len(late) == 1 and len(waited) == 2
    for r in waited:
        assert not r["ok"] and "peer transport" in r.get("sns_error", "") and "gave up waiting" in r["sns_error"], r
        assert r["seconds"] < 60.0


@pytest.mark.parametrize("nranks,halo_nodes", [(2, 5776), (3, 64), (3, 5776)])
def test_protocol_selftest_between_concurrent_queues(nranks, halo_nodes):
    """sns_peer_selftest: the put / flag / wait protocol between nranks concurrently running HIP queues of one process (a ring of
    halo links; 5776 nodes = the 76 x 76 interface of the 10 M-tet duct's x-slabs), every payload of every round verified --
    including the double-buffered receive buffers and slot tables over 200 rounds -- then 200 timed rounds of each collective alone
    (printed).  (Ranks = threads here: at most 3, one hardware queue each; the process tests above have no such limit.)"""
    import ctypes as C
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    from stabilized_navier_stokes_flow_fenicsx_amd import _lib
    lib = _lib.load()
    us = (C.c_double * 3)()
    rc = lib.sns_peer_selftest(0, nranks, halo_nodes, 200, us)
    assert rc == 0, lib.sns_last_error().decode()
    print(f"  {nranks} ranks, {halo_nodes} halo nodes ({32 * halo_nodes / 1e3:.0f} kB) per link: exchange {us[0]:.1f} us, all-reduce {us[1]:.1f} us, "
          f"all-gather {us[2]:.1f} us per round")
    assert 0.0 < us[0] < 5000.0 and 0.0 < us[1] < 5000.0 and 0.0 < us[2] < 5000.0


def test_bench_py_runs_two_ranks_end_to_end_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2 --transport peer --shared-gpu`: the launcher, two rank processes on cuda:0 bootstrapped over gloo, the
    slab partition, the timed Newton steps over peer windows, the weak leg and ONE JSON line -- the whole multi-rank flow of the
    bench on a 1-GPU box (RCCL cannot do that).  The line says it is a rehearsal; its timings mean nothing."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SNS_BENCH_LOG_DIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--transport", "peer", "--shared-gpu", "--steps", "2",
                        "--warmup", "1", "--cells", "64,16,16", "--no-f64-rerun", "--weak-timeout", "200", "--launch-timeout", "400"],
                       capture_output=True, text=True, env=env, timeout=460)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    print(f"  {d['value']} {d['unit']}, {d['ms_per_step']} ms per step, its {[b for _, b, _ in d['config']['newton_log_fnorm_kspits_reason']]}, "
          f"weak leg {d['weak_scaling']}")
    assert d["n_gpus"] == 2 and d["transport"] == "peer" and d["shared_gpu_rehearsal"] is True and d["degraded"] is False
    assert d["scaling"] == "strong" and "x-slabs" in d["config"]["parallelism"]
    # (one Newton iteration per step, snes_max_it = 1: the SNES reason is "max_it" by construction; the Krylov solves must have run)
    assert all(b > 0 and a == a for a, b, _ in d["config"]["newton_log_fnorm_kspits_reason"])
    assert "bitwise" in d["halo_overlap_selfcheck"]
    assert d["weak_scaling"] and "value" in d["weak_scaling"] and d["peer_transport"] is None
    assert d["config"]["krylov_loop_last_solve"]["halo_exchanges"] > 0
