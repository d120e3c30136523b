"""One rank of tests/test_gpu_peer.py: a process of its own with its own HIP context, attached to the other ranks through the
peer-window transport (sns_peer_*).  All ranks of a test share the one GPU of the box -- HIP IPC, the put / flag / wait protocol
and the whole partitioned solver run exactly as they would with one GPU per rank; only the xGMI hop is missing.

    python tests/peer_worker.py RANK WORLD PORT KIND OUT.json
"""
import json
import os
import sys
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def main():
    rank, world, port, kind, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch
    import torch.distributed as dist
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT
    from stabilized_navier_stokes_flow_fenicsx_amd._lib import SnsError
    from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, PeerGroup
    out = dict(rank=rank, ok=False)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    peers = None
    try:
        torch.cuda.set_device(0)
        peers = PeerGroup(device="cuda:0", window_bytes=16 << 20, check_rounds=int(os.environ.get("SNS_TEST_CHECK_ROUNDS", "50")))
        kw = {}
        kw_env = json.loads(os.environ.get("SNS_TEST_OPTS", "{}"))          # (tests: option sets of the partitioned path)
        if kind.startswith("duct"):
            m = M.duct_mesh((24, 6, 6), 4.0, jitter=0.1)
            bcs = B.duct_bcs(m)
            if kind.endswith("-rep-dense"):                  # replicated tail of the hierarchy: the all-gather path
                kw = dict(amg_replicate_rows=1 << 20)
        else:                                                # RCB blocks with corner ghosts: up to 3 neighbours per rank
            m = M.cavity_mesh(12, jitter=0.1)
            bcs = B.cavity_bcs(m)
        mask, g = bcs.flatten()
        Re = 12.0
        # the serial answer, computed by every rank for itself
        Ps = FlowProblem(m, (mask, g), reynolds=Re, **kw)
        Us, rs = Ps.stokes_solve()
        ws, ns = Ps.newton_solve(Us.clone())
        xs = torch.from_numpy(np.random.default_rng(3).normal(size=m.num_dofs)).cuda()
        Ps.jacobian(ws, "ns")
        ys = Ps.spmv(xs).cpu().numpy()
        Us, ws = Us.cpu().numpy(), ws.cpu().numpy()
        Ps.close()

        kw.update(kw_env)
        P = FlowProblem.distributed(m, (mask, g), group=peers, reynolds=Re, **kw)
        info = P.comm_info()
        part = P.part
        t0 = time.time()
        if kind == "duct-late" and rank == 1:
            # this rank never joins the solve: the others must give up with SNS_E_COMM, not hang
            time.sleep(float(os.environ.get("SNS_PEER_TIMEOUT_MS", "20000")) / 1000.0 + 3.0)
            out.update(ok=True, skipped=True)
        else:
            try:
                U, r = P.stokes_solve()
                w, n = P.newton_solve(U.clone())
                P.jacobian(w, "ns")
                y = P.spmv(P.scatter(xs.cpu().numpy()))
                Ug, wg, yg = P.gather(U).cpu().numpy(), P.gather(w).cpu().numpy(), P.gather(y).cpu().numpy()
                c = P.counters()
                out.update(ok=True, transport=info["transport"], nranks=info["nranks"], stokes_its=r.its, stokes_reason=r.reason,
                           newton_its=n.its, newton_reason=n.reason, ksp_its=list(n.ksp_its) if hasattr(n.ksp_its, "__iter__") else n.ksp_its,
                           serial=dict(stokes_its=rs.its, newton_its=ns.its, newton_reason=ns.reason),
                           err_stokes=rel(Ug, Us), err_newton=rel(wg, ws), err_spmv=rel(yg, ys), stokes_rnorm=r.rnorm,
                           levels=P.timings().amg_levels, cycle=[(x["kind"], x["pre"], x["post"]) for x in P.cycle()], exchanges=c.get("exchanges"), allreduces=c.get("allreduces"),
                           n_owned=int(part.n_owned), neighbors=[int(v) for v in part.neighbors], seconds=time.time() - t0)
            except SnsError as e:
                out.update(ok=False, sns_error=str(e), seconds=time.time() - t0)
        if kind != "duct-late":
            P.close()
        # ("duct-late": the ranks that gave up may have collectives of the dead solve in flight; the process ends instead)
    except BaseException as e:      # noqa: BLE001
        out.update(ok=False, error="".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:])
    with open(out_path, "w") as f:
        json.dump(out, f)
    try:
        if peers is not None and kind != "duct-late":
            peers.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
