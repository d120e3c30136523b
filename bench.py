#!/usr/bin/env python
"""Headline benchmark: Newton iterations of the stabilised P1-P1 Navier-Stokes solve
on the 10.1 M-tet square duct (BASELINE.json configs[4], the configuration the metric
is quoted on; it fits one MI355X), Re = 200.

A "step" is ONE Newton iteration of a real Newton sequence started from the Stokes
solution: fused Jacobian+residual assembly (HIP), AMG setup, FGMRES solve to rtol 1e-8
(the reference's KSP tolerance, NavierStokesChannelFlow.py:283), bt line-search residual.
When the sequence converges (||F|| < 1e-8, :281) it restarts from the Stokes solution.
metric = M-DOF/s = N_dof / (t_assemble + t_solve) per Newton iteration / 1e6  (SURVEY 8d).

  python bench.py [--gpus N --steps K --warmup W]           (N>1 under torch.distributed.run)
N>1 (weak scaling): the duct is refined by N^(1/3) per direction, every GPU keeps a ~10.1 M-tet x-slab
(--strong: the SAME mesh is element-partitioned over the ranks instead), halo exchange
and dot-product all-reduces on RCCL inside libsns.so.
"""
import argparse
import glob
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL / cross-process device memory need this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # torch.distributed.run exports OMP_NUM_THREADS=1 to its workers; the host-side symbolic setup (BSR pattern,
    # gather lists: OpenMP in libsns.so) wants this rank's share of the cores.  Must happen before libgomp loads.
    _lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"]))
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, (os.cpu_count() or 8) // max(1, _lw))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(sample_cells=(140, 35, 35), re_full=200.0, full_ny=75):
    """The oracle's C/OpenMP restatement ("port", oracle/c) timed on the host cores on a bounded
    sample of the same workload: ONE Newton iteration (assemble J+F, solve to rtol 1e-8) on the
    ~1 M-tet duct at the same cell Reynolds number Re*h as the full run, with the REFERENCE's
    linear algorithm: KSP tfqmr (NavierStokesChannelFlow.py:77,282-283) + PETSc's default
    preconditioner, block-Jacobi (one block per thread) with ILU(0) on each block."""
    from oracle import cport
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
    m = M.duct_mesh(sample_cells, 4.0)
    mask, g = B.duct_bcs(m).flatten()
    Re = re_full * sample_cells[1] / full_ny
    # one GPU's host share is 16 cores on the bench box; the reference's own runs use 6 ranks (run_all_images.sh:6)
    cport.set_num_threads(min(16, os.cpu_count() or 1, cport.num_threads()))
    nthr = cport.num_threads()
    print(f"[bench] cpu_baseline: {m.num_tets} tets on {nthr} threads", file=sys.stderr, flush=True)
    rp, ci = cport.pattern(m.num_nodes, m.tets)
    vals, F0 = cport.assemble("stokes", m.points, m.tets, None, 1.0, mask, g, rp, ci)
    U, sits, sreason, _ = cport.solve(m.num_nodes, rp, ci, vals, -F0, method="tfqmr", pc="ilu0", rtol=1e-8, maxit=2000)
    print(f"[bench] cpu_baseline: Stokes presolve {sits} its reason {sreason}", file=sys.stderr, flush=True)
    t0 = time.time()
    vals, F = cport.assemble("ns", m.points, m.tets, U, Re, mask, g, rp, ci)
    t1 = time.time()
    y, its, reason, rn = cport.solve(m.num_nodes, rp, ci, vals, F, method="tfqmr", pc="ilu0", rtol=1e-8, maxit=3000)
    t2 = time.time()
    _, Fn = cport.assemble("ns", m.points, m.tets, U - y, Re, mask, g, rp, ci)      # line-search residual
    t3 = time.time()
    ndof = m.num_dofs
    return {"value": round(ndof / (t3 - t0) / 1e6, 4), "unit": "M-DOF/s", "cores": nthr, "kind": "port",
            "sample": f"1 Newton iteration on duct {sample_cells} = {m.num_tets} tets / {ndof} dofs at Re={Re:.1f} "
                      f"(same Re*h as the full run): C/OpenMP assembly {t1 - t0:.2f}s + tfqmr/bjacobi({nthr})-ILU(0) "
                      f"{t2 - t1:.2f}s ({its} its, reason {reason}) + residual {t3 - t2:.2f}s; "
                      f"||F|| {np.linalg.norm(F):.2e} -> {np.linalg.norm(Fn):.2e}"}


def pmc_traffic(kernel_substr="k_spmv<2, 1"):
    """Per-launch HBM bytes of the dominant kernel from the committed rocprofv3 --pmc CSVs
    (profiles/*pmc*counter_collection.csv), corrected as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE is in KiB and reads half the bytes of a wide streaming read on gfx950 (x2);
    WRITE_SIZE (KiB) is exact.  None if no such profile is committed."""
    import csv
    fetch = write = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*counter_collection.csv"))):   # latest round wins
        tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") in tot:
                    tot[row["Counter_Name"]][0] += float(row["Counter_Value"])
                    tot[row["Counter_Name"]][1] += 1
        if tot["FETCH_SIZE"][1]:
            fetch = tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1] * 1024.0 * 2.0
        if tot["WRITE_SIZE"][1]:
            write = tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1] * 1024.0
    if fetch is None or write is None:
        return None
    return fetch + write

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=str, default="300,75,75")
    ap.add_argument("--length", type=float, default=4.0, help="duct length of the single-GPU share")
    ap.add_argument("--re", type=float, default=200.0)
    ap.add_argument("--ksp", type=str, default="bicgstab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64-rerun", action="store_true", help="skip the all-fp64 repetition of the timed steps")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="extra sns_options field for experiments, e.g. --opt amg_agg_size=4")
    ap.add_argument("--strong", action="store_true",
                    help="N>1: partition the SAME --cells mesh over the ranks instead of refining the duct with N")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M
    from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get("SNS_FORCE_DIST"))          # rehearse the partitioned path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29561")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    cells = tuple(int(c) for c in args.cells.split(","))
    opts = dict(reynolds=args.re, ksp_type=args.ksp, pc_type="amg", snes_max_it=1)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        opts[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
    weak = (world > 1 or force_dist) and not args.strong      # SNS_FORCE_DIST=1: rehearse the N>1 code path with one rank
    if weak:
        # weak scaling: the duct is refined uniformly so that every GPU keeps about the single-GPU share
        # (--cells tets): cells x N^(1/3) per direction, same geometry and Re.  Ranks own x-slabs; each rank
        # meshes only its own slab (+ one ghost cell layer), never the global mesh.
        from stabilized_navier_stokes_flow_fenicsx_amd import partition as PT
        sc = float(world) ** (1.0 / 3.0)
        cells = tuple(int(round(c * sc)) for c in cells)
        length = args.length
        part = PT.duct_slab_part(cells, length, rank, world)
        P = FlowProblem.from_part(part, device=f"cuda:{local_rank}", **opts)
        n_dof_global = 4 * (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
        n_tets_global = 6 * cells[0] * cells[1] * cells[2]
    else:
        length = args.length
        mesh = M.duct_mesh(cells, length)
        bcs = B.duct_bcs(mesh)
        if world > 1 or force_dist:
            P = FlowProblem.distributed(mesh, bcs, device=f"cuda:{local_rank}", **opts)
        else:
            P = FlowProblem(mesh, bcs, device=f"cuda:{local_rank}", **opts)
        n_dof_global = mesh.num_dofs
        n_tets_global = mesh.num_tets
    U, sres = P.stokes_solve()                       # initial guess, as the reference does (:519-523)
    if sres.reason <= 0:
        raise RuntimeError(f"Stokes solve did not converge: {sres}")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    w = U.clone()
    seq = 0
    log = []

    def step():
        nonlocal w, seq
        w, r = P.newton_solve(w)
        seq += 1
        log.append((r.fnorms[-1] if r.fnorms else float("nan"), r.ksp_its, r.reason))
        if r.reason == 2 or r.reason == 3 or seq >= 30:       # sequence converged: start over
            w = U.clone()
            seq = 0
        return r

    for _ in range(args.warmup):
        step()
    P.reset_timings()
    P.time_kernels(True)
    log.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    P.time_kernels(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / max(1, args.steps) * 1e3
    value = n_dof_global / (ms_per_step * 1e-3) / 1e6

    tm = P.timings()
    kt = P.kernel_times()
    s = P.sizes()
    # K1 (Jacobian + residual assembly) timed on its own after the timed region: HIP events around 5 passes
    asm_ms = P.bench_assemble(w, "ns", 5)
    asm_bytes = 2480.0 * s["n_tets"]                   # 2480 B/tet
    # the same K steps once more with EVERY array in fp64 (no fp32 copies inside the preconditioner), reported
    # beside the headline so that the effect of the mixed-precision preconditioner is on record
    all_f64 = None
    if P.options.amg_f32_matrix and not args.no_f64_rerun:
        main_log = list(log)
        P.set_options(amg_f32_matrix=0)
        w, seq = U.clone(), 0
        for _ in range(args.warmup):
            step()
        log.clear()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt64 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt64], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt64 = float(t)
        ms64 = dt64 / max(1, args.steps) * 1e3
        all_f64 = {"value": round(n_dof_global / (ms64 * 1e-3) / 1e6, 3), "unit": "M-DOF/s", "ms_per_step": round(ms64, 3),
                   "ksp_its": [b for _, b, _ in log]}
        P.set_options(amg_f32_matrix=1)
        log[:] = main_log
    # dominant kernel: the level-0 block-Jacobi sweep of the AMG cycle (3 of the 5 fine-level matrix passes
    # per preconditioner application).  Algorithmic bytes per launch:
    #   per nonzero block: values (64 B as the preconditioner's fp32 copy, 128 B in fp64) + 4 B column index
    #   per block row: 4 rowptr + 32 x + 32 b + 128 Dinv + 32 y = 228 B
    f32 = bool(P.options.amg_f32_matrix)
    kname = "k_spmv_f32<SPMV_JACOBI,FINE>" if f32 else "k_spmv<SPMV_JACOBI,FINE>"
    jac_ms, jac_calls = kt["jacobi"]
    alg_bytes = (68.0 if f32 else 132.0) * s["nnzb"] + 228.0 * s["n_owned"]
    roofline = None
    if jac_calls > 0:
        avg_ms = jac_ms / jac_calls
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic("k_spmv_f32<2, 1" if f32 else "k_spmv<2, 1"),
                    "kernel": kname, "avg_launch_ms": round(avg_ms, 5),
                    "launches": int(jac_calls), "algorithmic_bytes_per_launch": alg_bytes,
                    "assembly_kernels": {"avg_ms": round(asm_ms, 4), "algorithmic_bytes": asm_bytes,
                                         "achieved": round(asm_bytes / (asm_ms * 1e-3) / 1e9, 1),
                                         "frac": round(asm_bytes / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "path": "scratch-free (k_fused_offdiag + k_fused_diag)"
                                                 if P.options.assembly_fused else "staged (k_element + gathers)",
                                         # secondary: executed fp64 VALU flops of the scratch-free path,
                                         # counted from the gfx950 ISA: 300 fmac + 82 fma + 244 mul + 97 add per
                                         # block contribution = 1105 flop, 16 contributions per tet
                                         "fp64_vector": ({"executed_flops_per_tet": 17680,
                                                          "achieved_tflops": round(17680.0 * s["n_tets"] / (asm_ms * 1e-3) / 1e12, 2),
                                                          "peak_tflops": 78.6,
                                                          "frac": round(17680.0 * s["n_tets"] / (asm_ms * 1e-3) / 1e12 / 78.6, 4)}
                                                         if P.options.assembly_fused else None)},
                    "other_fine_spmv": {k: {"avg_ms": round(v[0] / v[1], 5), "launches": int(v[1])}
                                        for k, v in kt.items() if v[1] > 0 and k != "jacobi"}}
    out = {
        "metric": "M-DOF/s (assembly+solve) per Newton iteration",
        "value": round(value, 3), "unit": "M-DOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "precision_note": "operator, residuals, Krylov recurrences and reductions in f64; the AMG preconditioner's "
                          "smoother/residual passes read an fp32 copy of the level matrices (arithmetic f64)"
                          if P.options.amg_f32_matrix else "all f64",
        "config": {"workload": f"duct [0,{length:g}]x[-.5,.5]^2, {cells[0]}x{cells[1]}x{cells[2]} cells = {n_tets_global} tets, "
                               f"{n_dof_global} dofs, Re={args.re:g}, Newton iteration (assemble J+F, AMG setup, "
                               f"{args.ksp} rtol 1e-8, bt line search)",
                   "parallelism": (f"x-slab element partition x{world}, {n_tets_global // world} tets per GPU"
                                   if world > 1 else "single GPU"),
                   "scaling_note": ("weak: every GPU keeps BASELINE config 5's 10.1 M-tet share (duct refined by "
                                    "N^(1/3)); --strong splits the one 10.1 M-tet mesh N ways (DESIGN.md section 7)"
                                    if weak else ("strong: the one mesh split N ways" if world > 1 else
                                                  "N=1 is BASELINE config 5's mesh on one GPU")),
                   "newton_log_fnorm_kspits_reason": [(float(f"{a:.3e}"), b, c) for a, b, c in log],
                   "phase_ms_per_step": {"assemble": round(tm.assemble_ms / args.steps, 3),
                                         "pc_setup": round(tm.pc_setup_ms / args.steps, 3),
                                         "krylov": round(tm.krylov_ms / args.steps, 3)},
                   "amg_levels": tm.amg_levels, "stokes_its": sres.its},
        "roofline": roofline,
        "all_f64_preconditioner": all_f64,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(re_full=args.re, full_ny=cells[1])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    P.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
