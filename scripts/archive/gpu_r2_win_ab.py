"""round 2: interleaved A/B of the low-precision Jacobi sweep, fp16 row-scaled vs fp32 (sns_bench_variants which=1),
on the headline operator (10.1 M-tet duct Jacobian).  Run with SNS_BOTH_LP=1."""
import ctypes as C, os, sys, torch
os.environ["SNS_BOTH_LP"] = "1"
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
from stabilized_navier_stokes_flow_fenicsx_amd._lib import check
m = M.duct_mesh((300, 75, 75), 4.0)
P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, amg_f32_matrix=2)
U, _ = P.stokes_solve()
P.jacobian(U, "ns")
P.pc_setup()
out = (C.c_double * 2)()
check(P.lib.sns_bench_variants(P.h, 1, 6, 10, out))
s = P.sizes()
for nm, ms, bpb, extra in (("fp16", out[0], 36.0, 16.0), ("fp32", out[1], 68.0, 0.0)):
    alg = bpb * s["nnzb"] + (228.0 + extra) * s["n_owned"]
    print(f"Jacobi sweep {nm}: {ms:.4f} ms, algorithmic {alg/1e9:.3f} GB -> {alg/ms/1e6:.0f} GB/s = {alg/ms/1e6/8000:.3f} of HBM peak", flush=True)
P.close()
