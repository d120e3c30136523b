"""round 3: residual history of the jittered 648 k-tet duct at Re 200 (the damping-retry test case) under the automatic damping"""
import sys, io, os, re
sys.path.insert(0, ".")
import torch
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
m = M.duct_mesh((120, 30, 30), 4.0, jitter=0.2)
for kw in (dict(), dict(amg_fused_post=0), dict(amg_nu_l1_pre=4, amg_nu_l1_post=4), dict(amg_fused_post=0, amg_nu_l1_pre=4, amg_nu_l1_post=4)):
    P = FlowProblem(m, B.duct_bcs(m), reynolds=200.0, ksp_max_it=600, amg_retry_damping=0, **kw)
    U, r = P.stokes_solve()
    F = P.zeros()
    P.jacobian(U, "ns", residual_out=F)
    P.set_options(monitor=1)
    sys.stdout.flush()
    y, k = P.krylov_solve(F)
    P.set_options(monitor=0)
    print("OPTS", kw, "stokes its", r.its, "-> its", k.its, "reason", k.reason, "rnorm", k.rnorm, "|F|", float(F.norm()), flush=True)
    P.close()
