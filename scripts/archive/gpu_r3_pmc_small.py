"""Small driver for single-counter PMC passes (round 3): a duct Jacobian (cells from argv, default 1 M tets), a few fp64 SpMVs and V-cycles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
cells = tuple(int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "140,35,35").split(","))
m = M.duct_mesh(cells, 4.0)
P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0)
U, _ = P.stokes_solve()
P.jacobian(U, "ns", residual_out=P.zeros())
P.pc_setup()
x = torch.randn_like(U)
y = P.zeros()
for _ in range(4):
    P.spmv(x, y)
for _ in range(4):
    P.pc_apply(x, y)
torch.cuda.synchronize()
print("done", flush=True)
