"""Can two RCCL ranks share ONE GPU on this pool?  (would allow rehearsing the N > 1 transport on the 1-GPU box)"""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
r = dist.get_rank()
x = torch.full((4,), float(r + 1), device="cuda", dtype=torch.float64)
dist.all_reduce(x)
y = torch.zeros(4, device="cuda", dtype=torch.float64)
if r == 0:
    dist.send(x, 1); dist.recv(y, 1)
else:
    dist.recv(y, 0); dist.send(x, 0)
torch.cuda.synchronize()
print("rank", r, "allreduce", x.tolist(), "p2p", y.tolist(), flush=True)
dist.destroy_process_group()
