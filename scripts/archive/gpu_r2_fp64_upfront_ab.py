"""round 2: fp64 y = Ax, production (the first 16 blocks of a row requested up-front) vs the stepped loop it replaced
(sns_bench_variants which=3), interleaved in one process."""
import sys, os, ctypes as C
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
from stabilized_navier_stokes_flow_fenicsx_amd._lib import check
for cells in [(300, 75, 75), (150, 38, 38)]:
    m = M.duct_mesh(cells, 4.0)
    P = FlowProblem(m, B.duct_bcs(m), reynolds=100.0)
    U, r = P.stokes_solve()
    P.jacobian(U, "ns"); P.pc_setup()
    ms = (C.c_double * 2)()
    check(P.lib.sns_bench_variants(P.h, 3, 6, 10, ms))
    print(cells, "fp64 y=Ax: up-front (production) %.4f ms  stepped loop %.4f ms  ratio %.3f" % (ms[0], ms[1], ms[0] / ms[1]), flush=True)
    P.close()
