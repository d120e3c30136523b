"""Round 3: what an AMG setup (sns_pc_setup) of the 10 M-tet Jacobian consists of -- run under rocprofv3 --kernel-trace --stats:
16 setups after the first Stokes solve + Jacobian (every 4th re-estimates the spectra), wall time per setup printed."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.duct_mesh((300, 75, 75), 4.0)
P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0)
U, _ = P.stokes_solve()
F = P.zeros(); P.jacobian(U, "ns", residual_out=F)
P.pc_setup(); torch.cuda.synchronize()
ts = []
for i in range(16):
    t0 = time.time(); P.pc_setup(); torch.cuda.synchronize(); ts.append(1e3 * (time.time() - t0))
print("setup ms:", [round(t, 2) for t in ts], "mean", round(sum(ts) / len(ts), 2))
P.close()
