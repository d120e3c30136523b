"""Round 3 probe: does this RCCL accept two ranks on ONE device (so that the RCCL transport of libsns.so could be rehearsed at N = 2 on a
one-GPU box)?  RESULT (round 3): no -- both ranks exit with code 1 at communicator creation (duplicate device), before the
first collective; the RCCL transport at N > 1 stays covered by the single-rank RCCL test, the team-transport tests (same exchange plans, same
two-stream choreography) and the gloo dry run only.  Run as: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/gpu_r3_rccl_same_device.py"""
import os, sys, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
r = dist.get_rank()
try:
    x = torch.full((4,), float(r + 1), device="cuda:0")
    dist.all_reduce(x); torch.cuda.synchronize()
    print(f"rank {r}: all_reduce ok {x.tolist()}", flush=True)
    y = torch.zeros(4, device="cuda:0")
    ops = [dist.P2POp(dist.isend, x, 1 - r), dist.P2POp(dist.irecv, y, 1 - r)]
    for w in dist.batch_isend_irecv(ops): w.wait()
    torch.cuda.synchronize(); print(f"rank {r}: sendrecv ok {y.tolist()}", flush=True)
except Exception as e:
    print(f"rank {r}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)
dist.destroy_process_group()
