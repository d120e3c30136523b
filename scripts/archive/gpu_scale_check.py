"""Scale check: timings of setup / assembly / SpMV / solves on growing duct meshes."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabilized_navier_stokes_flow_fenicsx_amd import mesh as M, bcs as B
from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem

cases = [eval(a) for a in sys.argv[1:]] or [(100, 25, 25)]
for cells in cases:
    t0 = time.time(); m = M.duct_mesh(cells, 4.0); t1 = time.time()
    mask, g = B.duct_bcs(m).flatten(); t2 = time.time()
    print(f"== cells {cells}: {m.num_tets} tets {m.num_nodes} nodes; mesh {t1-t0:.1f}s bcs {t2-t1:.1f}s", flush=True)
    P = FlowProblem(m, (mask, g), reynolds=200.0, pc_type="amg", ksp_type="fgmres", gmres_restart=30)
    t3 = time.time(); s = P.sizes()
    print(f"   create {t3-t2:.1f}s nnzb {s['nnzb']}", flush=True)
    U, res = P.stokes_solve(); t4 = time.time()
    tm = P.timings()
    print(f"   stokes: {res} wall {t4-t3:.2f}s (asm {tm.assemble_ms:.1f} pc {tm.pc_setup_ms:.1f} krylov {tm.krylov_ms:.1f} ms, levels {tm.amg_levels})", flush=True)
    nnzb, n = s["nnzb"], s["n_owned"]
    ms = P.bench_spmv(20)
    by = 132.0 * nnzb + 68.0 * n
    print(f"   spmv {ms:.4f} ms  -> {by/ms/1e6:.1f} GB/s algorithmic", flush=True)
    ms = P.bench_assemble(U, "ns", 3)
    print(f"   assemble(J+F) {ms:.3f} ms -> {2480.0*s['n_tets']/ms/1e6:.1f} GB/s algorithmic", flush=True)
    P.reset_timings()
    for ksp in ("fgmres", "bicgstab"):
        P.set_options(ksp_type=ksp, snes_max_it=1)
        w = U.clone(); t5 = time.time()
        w, r = P.newton_solve(w); t6 = time.time()
        tm = P.timings()
        print(f"   newton[1 it] {ksp}: its {r.its} reason {r.reason} ksp_its {r.ksp_its} fnorms {['%.2e' % f for f in r.fnorms]} wall {t6-t5:.2f}s (asm {tm.assemble_ms:.1f} pc {tm.pc_setup_ms:.1f} krylov {tm.krylov_ms:.1f} ms)", flush=True)
        P.reset_timings()
    P.close(); del P
