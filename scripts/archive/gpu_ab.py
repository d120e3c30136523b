"""Interleaved A/B of kernel variants in ONE process (boxes differ by +-10 %, so cross-run comparisons are useless)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
from stabilized_navier_stokes_flow_fenicsx_amd._lib import check
for cells in [(300, 75, 75), (150, 38, 38)]:
    m = M.duct_mesh(cells, 4.0)
    P = FlowProblem(m, B.duct_bcs(m), reynolds=100.0)
    U, r = P.stokes_solve()
    P.jacobian(U, "ns"); P.pc_setup()
    for which, name in ((0, "fp64 y=Ax: default vs nt"), (1, "f32 Jacobi: production vs variant 1"), (2, "f32 Jacobi: production vs variant 2"),
                        (3, "fp64 y=Ax nt: cooperative loads vs r1e loop")):
        ms = (C.c_double * 2)()
        check(P.lib.sns_bench_variants(P.h, which, 6, 10, ms))
        print(cells, name, "A %.4f ms  B %.4f ms  ratio %.3f" % (ms[0], ms[1], ms[1] / ms[0]), flush=True)
    P.close()
