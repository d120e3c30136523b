"""round 4: bench.py's STRONG layout over the peer-window transport -- N PROCESSES (one HIP context each, windows mapped through
HIP IPC) on the ONE GPU of the box; x-slabs of the 300x75x75 duct as in scripts/gpu_r4_strong_rehearsal.py (which runs the same
split as threads over the emulated team transport).

Reported per N: iteration counts and collective counters (must equal the team rehearsal's), the protocol cost of each collective
back to back (sns_bench_collective: launches + flag round trips with all ranks' kernels on one GPU -- no xGMI hop, and the ranks
compete for the same CUs, so an upper bound of the launch part and no statement about the link), and the wall time of the step
(not meaningful as a scaling figure: N ranks share one GPU).

usage: python scripts/gpu_r4_peer_strong.py 2,4 [cells] [KEY=VALUE ...]          (parent)
       python scripts/gpu_r4_peer_strong.py --rank R N PORT cells OUT [KEY=VALUE ...]   (one rank, started by the parent)
"""
import json
import os
import socket
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(argv):
    rank, N, port, cells, out_path = int(argv[0]), int(argv[1]), int(argv[2]), tuple(int(c) for c in argv[3].split(",")), argv[4]
    opts = {}
    for a in argv[5:]:
        k, v = a.split("=")
        opts[k] = float(v) if "." in v else int(v)
    import torch
    import torch.distributed as dist
    from stabilized_navier_stokes_flow_fenicsx_amd import partition as PT
    from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, PeerGroup
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=N)
    torch.cuda.set_device(0)
    peers = PeerGroup(device="cuda:0")
    part = PT.duct_slab_part(cells, 4.0, rank, N)
    P = FlowProblem.from_part(part, group=peers, reynolds=200.0, snes_max_it=1, **opts)
    U, r = P.stokes_solve()
    w, n1 = P.newton_solve(U.clone())
    c = P.counters()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    w, n2 = P.newton_solve(w)
    torch.cuda.synchronize()
    step_s = time.time() - t0
    tm = P.timings()
    lat = {}
    for which, count in (("exchange", 0), ("allreduce", 5), ("allgather", 2048)):
        dist.barrier()
        lat[which] = 1e3 * P.bench_collective(which, count=count, reps=300)          # us
    out = dict(rank=rank, n_owned=int(part.n_owned), neighbors=[int(v) for v in part.neighbors], halo_nodes=int(len(part.recv_idx)),
               stokes_its=r.its, ksp_its=[n1.ksp_its, n2.ksp_its], fnorm=float(n2.fnorms[-1]), rows=[h["rows"] for h in P.hierarchy()],
               cycle=[(x["kind"], x["pre"], x["post"]) for x in P.cycle()], allreduces=c["allreduces"], exchanges=c["exchanges"],
               step_ms=1e3 * step_s, krylov_ms=getattr(tm, "krylov_ms", None), latency_us=lat, transport=P.comm_info()["transport"])
    P.close()
    peers.close()
    with open(out_path, "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "--rank":
        return rank_main(sys.argv[2:])
    args = [a for a in sys.argv[1:] if "=" not in a]
    kv = [a for a in sys.argv[1:] if "=" in a]
    cells = "300,75,75" if len(args) < 2 else args[1]
    os.makedirs("gpurun_out", exist_ok=True)
    print("options", kv, "cells", cells, flush=True)
    for N in [int(a) for a in args[0].split(",")]:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        outs = [f"gpurun_out/peer_strong_N{N}_rank{r}.json" for r in range(N)]
        for o in outs:
            if os.path.exists(o):
                os.remove(o)
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(N), str(port), cells, outs[r]] + kv)
                 for r in range(N)]
        try:
            for p in procs:
                p.wait(timeout=max(1.0, 900.0 - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            print(f"N={N}: ranks did not finish within 900 s", flush=True)
            return 1
        if any(p.returncode for p in procs) or not all(os.path.exists(o) for o in outs):
            print(f"N={N}: exit codes {[p.returncode for p in procs]}", flush=True)
            return 1
        res = [json.load(open(o)) for o in outs]
        o = res[0]
        mid = res[min(1, N - 1)]
        its = o["ksp_its"][0]
        print(f"N={N} ({o['transport']}): stokes its {o['stokes_its']} newton ksp its {o['ksp_its']} |F| {o['fnorm']:.2e} rows(rank 0) {o['rows']} "
              f"cycle {o['cycle']}; first Newton solve: {o['allreduces']} all-reduces, {o['exchanges']} halo exchanges = "
              f"{o['allreduces'] / max(1, its):.1f} / {o['exchanges'] / max(1, its):.1f} per iteration", flush=True)
        print(f"      same decisions on every rank: {all((r['stokes_its'], r['ksp_its']) == (o['stokes_its'], o['ksp_its']) for r in res)}; "
              f"rank {mid['rank']}: {mid['n_owned']} nodes, neighbours {mid['neighbors']}, {mid['halo_nodes']} halo nodes "
              f"({32 * mid['halo_nodes'] / 1e3:.0f} kB per exchange); back-to-back collectives, us each (max over ranks): "
              + ", ".join(f"{k} {max(r['latency_us'][k] for r in res):.1f}" for k in ("exchange", "allreduce", "allgather"))
              + f"; step wall {max(r['step_ms'] for r in res):.1f} ms with {N} ranks on one GPU (wall {time.time() - t0:.0f} s)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
