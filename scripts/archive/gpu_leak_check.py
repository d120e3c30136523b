"""Create / solve / destroy repeatedly and watch the device memory (hipMalloc leaks show up as a drift)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B, partition as PT
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, Team
m = M.duct_mesh((60, 15, 15), 4.0)
bc = B.duct_bcs(m)
free0 = None
for it in range(25):
    P = FlowProblem(m, bc, reynolds=40.0)
    U, r = P.stokes_solve()
    w, n = P.newton_solve(U.clone())
    P.set_options(ksp_type="fgmres"); P.newton_solve(U.clone())
    P.close()
    del P, U, w
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2: free0 = free
    if it % 6 == 0 or it == 24: print(f"iter {it}: free {free / 2**20:.0f} MiB", flush=True)
print("serial drift MiB:", (free0 - free) / 2**20)
f1 = None
for it in range(6):
    team = Team(4)
    def work(rank, team):
        P = FlowProblem.from_part(PT.duct_slab_part((60, 15, 15), 4.0, rank, 4), group=team, reynolds=40.0, amg_coarse_size=24)
        U, r = P.stokes_solve(); w, n = P.newton_solve(U.clone()); P.close(); return n.reason
    team.run(work); team.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 1: f1 = free
print("team drift MiB:", (f1 - free) / 2**20)
