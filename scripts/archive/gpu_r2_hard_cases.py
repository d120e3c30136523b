"""round 2: the cases where BiCGStab + AMG fails at the Stokes guess (cell Reynolds number 5-10): other Krylov methods, more
smoothing, stronger damping, Reynolds continuation."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem, newton_with_reynolds_continuation
cases = [("jittered duct 120x30x30 Re 200", lambda: M.duct_mesh((120, 30, 30), 4.0, jitter=0.2), 200.0),
         ("duct 200x50x50 Re 500", lambda: M.duct_mesh((200, 50, 50), 4.0), 500.0)]
variants = [dict(ksp_type="bicgstab"), dict(ksp_type="fgmres", gmres_restart=30), dict(ksp_type="fgmres", gmres_restart=100),
            dict(ksp_type="tfqmr"), dict(ksp_type="bicgstab", amg_nu=2), dict(ksp_type="bicgstab", amg_omega=0.6),
            dict(ksp_type="fgmres", gmres_restart=100, amg_nu=2)]
for name, make, Re in cases:
    m = make(); bcs = B.duct_bcs(m)
    for v in variants:
        P = FlowProblem(m, bcs, reynolds=Re, ksp_max_it=1500, **v)
        U, r = P.stokes_solve()
        t = time.time(); w, n = P.newton_solve(U.clone()); torch.cuda.synchronize(); dt = time.time() - t
        print(f"{name} {v}: stokes {r.its}/{r.reason} newton {n.its}/{n.reason} ksp {n.ksp_its} |F| {n.fnorms[-1]:.2e} {dt:.2f}s", flush=True)
        P.close()
    P = FlowProblem(m, bcs, reynolds=Re, ksp_max_it=1500)
    U, r = P.stokes_solve()
    t = time.time(); w, n = newton_with_reynolds_continuation(P, U.clone(), verbose=True); dt = time.time() - t
    print(f"{name} continuation: newton {n.its}/{n.reason} ksp {n.ksp_its} {dt:.2f}s", flush=True)
    P.close()
