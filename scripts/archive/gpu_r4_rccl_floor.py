"""Round 4: the launch floor of the RCCL collectives the partitioned solver issues per BiCGStab iteration, measured with ONE rank
(a 1-GPU box cannot host two): ncclAllReduce of 5 doubles and a send/recv group to self of one halo message (76 x 76 nodes x 4
doubles = 185 kB, the x-slab interface of the 10 M-tet duct), stream-ordered back to back and interleaved with a small kernel.
What a real link adds on top (xGMI hop latency, the peer's arrival time) is NOT in these numbers."""
import os, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
small = torch.ones(5, dtype=torch.float64, device="cuda")
halo_s = torch.ones(76 * 76 * 4, dtype=torch.float64, device="cuda")
halo_r = torch.empty_like(halo_s)
work = torch.ones(1 << 16, dtype=torch.float64, device="cuda")

def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

def sr():
    ops = [dist.P2POp(dist.isend, halo_s, 0), dist.P2POp(dist.irecv, halo_r, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()

k = timed(lambda: work.mul_(1.0))
ar = timed(lambda: dist.all_reduce(small))
ark = timed(lambda: (work.mul_(1.0), dist.all_reduce(small)))
srt = timed(sr)
srk = timed(lambda: (work.mul_(1.0), sr()))
ag = timed(lambda: dist.all_gather_into_tensor(halo_r[:5 * 1], small))
print(f"small kernel alone {k:.1f} us | all-reduce(5 doubles) {ar:.1f} us, with a kernel between {ark:.1f} us | "
      f"send/recv group to self (185 kB) {srt:.1f} us, with a kernel between {srk:.1f} us | all-gather(5) {ag:.1f} us")
dist.destroy_process_group()
