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
N>1: the HEADLINE is what north_star states -- the SAME 10.1 M-tet duct element-partitioned into N x-slabs (strong
scaling; every rank meshes only its own slab), halo exchange and dot-product all-reduces on RCCL inside libsns.so.
The weak layout (duct refined by N^(1/3) per direction, every GPU keeps a ~10.1 M-tet slab) is timed afterwards and
reported under "weak_scaling" in the same JSON line (--no-weak skips it).
--config 3 / 4 time the other full-size BASELINE configs (55^3 cavity Re 100; 240x60x60 two-stream channel) as
secondary lines; the default (5) is the headline.
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


def cpu_baseline(mesh, mask, g, U, Re, maxit=200):
    """The oracle's C/OpenMP restatement ("port", oracle/c) timed on the host cores on THE SAME workload as the GPU
    line (same mesh, BCs, initial guess, tolerances): ONE Newton iteration at the Stokes solution U
    of the full mesh -- assemble J+F, solve J y = F to rtol 1e-8, line-search residual -- with the REFERENCE's linear
    algorithm: KSP tfqmr (NavierStokesChannelFlow.py:77,282-283) + PETSc's default preconditioner in parallel,
    block-Jacobi (one block per thread) with ILU(0) on each block.  The Krylov solve is bounded by `maxit`
    iterations so that the default bench run stays within minutes; if it stops there the rate is an UPPER bound
    for the CPU and the sample text says so."""
    from oracle import cport
    # one GPU's host share is 16 cores on the bench box; the reference's own runs use 6 ranks (run_all_images.sh:6)
    cport.set_num_threads(min(16, os.cpu_count() or 1, cport.num_threads()))
    nthr = cport.num_threads()
    print(f"[bench] cpu_baseline: {mesh.num_tets} tets on {nthr} threads", file=sys.stderr, flush=True)
    rp, ci = cport.pattern(mesh.num_nodes, mesh.tets)
    t0 = time.time()
    vals, F = cport.assemble("ns", mesh.points, mesh.tets, U, Re, mask, g, rp, ci)
    t1 = time.time()
    print(f"[bench] cpu_baseline: assembly {t1 - t0:.1f}s", file=sys.stderr, flush=True)
    y, its, reason, rn = cport.solve(mesh.num_nodes, rp, ci, vals, F, method="tfqmr", pc="ilu0", rtol=1e-8, maxit=maxit)
    t2 = time.time()
    print(f"[bench] cpu_baseline: tfqmr {its} its reason {reason} in {t2 - t1:.1f}s", file=sys.stderr, flush=True)
    _, Fn = cport.assemble("ns", mesh.points, mesh.tets, U - y, Re, mask, g, rp, ci)      # line-search residual
    t3 = time.time()
    ndof = mesh.num_dofs
    f0, f1 = float(np.linalg.norm(F)), float(np.linalg.norm(Fn))
    bound = "" if reason > 0 else f" -- NOT converged within {maxit} iterations (||r||/||b|| {rn / f0:.1e}): upper bound"
    return {"value": round(ndof / (t3 - t0) / 1e6, 4), "unit": "M-DOF/s", "cores": nthr, "kind": "port",
            "t_asm_s": round(t1 - t0, 2), "t_solve_s": round(t2 - t1, 2), "t_residual_s": round(t3 - t2, 2),
            "ksp_its": its, "ksp_reason": reason,
            "sample": f"the GPU line's own workload: 1 Newton iteration at the Stokes solution on {mesh.num_tets} tets / "
                      f"{ndof} dofs, Re={Re:g}: C/OpenMP assembly {t1 - t0:.2f}s + tfqmr/bjacobi({nthr})-ILU(0) "
                      f"{t2 - t1:.2f}s ({its} its, reason {reason}) + residual {t3 - t2:.2f}s; "
                      f"||F|| {f0:.2e} -> {f1:.2e}{bound}"}


def pmc_traffic(kernel_substr):
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

def build_problem(cfg, cells, length, Re, world, rank, local_rank, opts, dist_on):
    """(P, n_dof_global, n_tets_global, description, host_inputs) of one BASELINE config on this rank.
    Config 5 (duct): x-slab element partition, every rank meshes only its own slab (partition.duct_slab_part).
    Configs 3 / 4: the global mesh is built on every rank and RCB-partitioned (setup cost, not timed)."""
    from stabilized_navier_stokes_flow_fenicsx_amd import bcs as B, mesh as M, partition as PT
    from stabilized_navier_stokes_flow_fenicsx_amd.solver import FlowProblem
    dev = f"cuda:{local_rank}"
    host = None
    if cfg == 5:
        if dist_on:
            part = PT.duct_slab_part(cells, length, rank, world)
            P = FlowProblem.from_part(part, device=dev, **opts)
        else:
            mesh = M.duct_mesh(cells, length)
            bcs = B.duct_bcs(mesh)
            P = FlowProblem(mesh, bcs, device=dev, **opts)
            host = (mesh, bcs)
        nd = 4 * (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
        nt = 6 * cells[0] * cells[1] * cells[2]
        desc = f"duct [0,{length:g}]x[-.5,.5]^2, {cells[0]}x{cells[1]}x{cells[2]} cells"
    else:
        if cfg == 3:                       # LidDrivenNavierStokesFlow.py extended to the unit cube
            mesh = M.cavity_mesh(cells[0])
            bcs = B.cavity_bcs(mesh)
            desc = f"lid-driven unit cube, {cells[0]}^3 cells"
        else:                              # NavierStokesChannelFlow.py two-stream inlet, ratio 0.5 (config 4)
            mesh = M.channel_mesh(cells)
            bcs = B.channel_bcs(mesh, *B.two_stream_profiles(0.5))
            desc = f"two-stream channel 4x1x1, {cells[0]}x{cells[1]}x{cells[2]} cells, flowrate ratio 0.5"
        P = FlowProblem.distributed(mesh, bcs, device=dev, **opts) if dist_on else FlowProblem(mesh, bcs, device=dev, **opts)
        host = (mesh, bcs)
        nd, nt = mesh.num_dofs, mesh.num_tets
    return P, nd, nt, desc, host


def timed_newton_steps(P, U, steps, warmup, world):
    """W untimed + K timed Newton iterations of a real sequence from the Stokes solution U (restarted when it has
    converged); barrier + synchronize on both sides, MAX over ranks.  Returns (ms_per_step, log)."""
    import torch
    import torch.distributed as dist

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    state = {"w": U.clone(), "seq": 0}
    log = []

    def step():
        w, r = P.newton_solve(state["w"])
        state["w"] = w
        state["seq"] += 1
        log.append((r.fnorms[-1] if r.fnorms else float("nan"), r.ksp_its, r.reason))
        if r.reason == 2 or r.reason == 3 or state["seq"] >= 30:       # sequence converged: start over
            state["w"] = U.clone()
            state["seq"] = 0

    for _ in range(warmup):
        step()
    P.reset_timings()
    P.time_kernels(True)
    log.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    P.time_kernels(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt / max(1, steps) * 1e3, log, state["w"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=5, choices=[3, 4, 5],
                    help="BASELINE config: 5 = 10.1 M-tet duct Re 200 (headline), 3 = 55^3 cavity Re 100, "
                         "4 = 240x60x60 two-stream channel Re 50")
    ap.add_argument("--cells", type=str, default=None)
    ap.add_argument("--length", type=float, default=4.0, help="duct length")
    ap.add_argument("--re", type=float, default=None)
    ap.add_argument("--ksp", type=str, default="bicgstab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-maxit", type=int, default=200, help="iteration bound of the CPU baseline's Krylov solve")
    ap.add_argument("--no-f64-rerun", action="store_true", help="skip the all-fp64 repetition of the timed steps")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the weak-scaling layout after the headline")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="extra sns_options field for experiments, e.g. --opt amg_agg_size=4")
    ap.add_argument("--strong", action="store_true", help="(default since round 2; kept for old command lines)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get("SNS_FORCE_DIST"))          # rehearse the partitioned path with one rank
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29561")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    cfg = args.config
    default_cells = {5: "300,75,75", 4: "240,60,60", 3: "55,55,55"}[cfg]
    cells = tuple(int(c) for c in (args.cells or default_cells).split(","))
    Re = args.re if args.re is not None else {5: 200.0, 4: 50.0, 3: 100.0}[cfg]
    opts = dict(reynolds=Re, ksp_type=args.ksp, pc_type="amg", snes_max_it=1)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        opts[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
    length = args.length

    # ---- headline: the SAME mesh on N GPUs (strong scaling; N = 1 is the mesh on one GPU) ----------------------
    P, n_dof_global, n_tets_global, desc, host = build_problem(cfg, cells, length, Re, world, rank, local_rank, opts, dist_on)
    U, sres = P.stokes_solve()                       # initial guess, as the reference does (:519-523)
    if sres.reason <= 0:
        raise RuntimeError(f"Stokes solve did not converge: {sres}")
    ms_per_step, log, w = timed_newton_steps(P, U, args.steps, args.warmup, world)
    value = n_dof_global / (ms_per_step * 1e-3) / 1e6

    tm = P.timings()
    kt = P.kernel_times()
    s = P.sizes()
    ctr = P.counters()
    # K1 (Jacobian + residual assembly) timed on its own after the timed region: HIP events around 5 passes
    asm_ms = P.bench_assemble(w, "ns", 5)
    asm_bytes = 2480.0 * s["n_tets"]                   # 2480 B/tet
    # the same K steps once more with EVERY array in fp64 (no fp32 copies inside the preconditioner), reported
    # beside the headline so that the effect of the mixed-precision preconditioner is on record
    all_f64 = None
    fmt0 = int(P.options.amg_f32_matrix)
    if fmt0 and not args.no_f64_rerun:
        P.set_options(amg_f32_matrix=0)
        ms64, log64, _ = timed_newton_steps(P, U, args.steps, args.warmup, world)
        all_f64 = {"value": round(n_dof_global / (ms64 * 1e-3) / 1e6, 3), "unit": "M-DOF/s", "ms_per_step": round(ms64, 3),
                   "ksp_its": [b for _, b, _ in log64]}
        P.set_options(amg_f32_matrix=fmt0)
    # The four fine-level matrix passes of a BiCGStab iteration (2 x Jacobi sweep + 2 x residual of the two V-cycles on the
    # preconditioner's matrix copy, y = Ax and y = Ax + <r^, y> on the fp64 operator) are 60 % of a step, each 22-27 % of
    # the SpMV time.  Algorithmic bytes per launch:
    #   per nonzero block: values + 4 B column index -- 128 B fp64 operator, 64 B fp32 copy, 32 B fp16 copy
    #   per block row:     4 rowptr + 32 per vector touched (x, b, y, dot weight) + D^-1 (Jacobi: 128 B fp64, 64 B as the
    #                      fp32 copy the low-precision sweeps read) + 16 row scales (fp16)
    fmt = int(P.options.amg_f32_matrix)
    lp = {0: ("k_spmv<{m}, 1, 1, 0>", 132.0, 0.0), 1: ("k_spmv_lp<{m}, 1, 0, 1, 1>", 68.0, 0.0), 2: ("k_spmv_lp<{m}, 1, 0, 2, 1>", 36.0, 16.0)}[fmt]
    nb, nr = float(s["nnzb"]), float(s["n_owned"])
    kinfo = {
        "jacobi": (lp[0].format(m=2), lp[1] * nb + (4 + 32 * 3 + (128 if fmt == 0 else 64) + lp[2]) * nr,
                   "AMG fine-level block-Jacobi sweep"),
        "b_minus_ax": (lp[0].format(m=1), lp[1] * nb + (4 + 32 * 3 + lp[2]) * nr, "AMG fine-level residual r = b - Ax"),
        "ax": ("k_spmv<0, 1, 1, 0>", 132.0 * nb + (4 + 32 * 2) * nr, "Krylov operator y = Ax (fp64)"),
        "ax_dot": ("k_spmv<3, 1, 1, 0>", 132.0 * nb + (4 + 32 * 3) * nr, "Krylov operator y = Ax + <r^, y> (fp64)"),
    }
    per_kernel = {}
    for key, (kname_k, bytes_k, what) in kinfo.items():
        ms_k, calls_k = kt[key]
        if calls_k > 0:
            avg = ms_k / calls_k
            ach = bytes_k / (avg * 1e-3) / 1e9
            per_kernel[key] = {"kernel": kname_k, "what": what, "avg_launch_ms": round(avg, 5), "launches": int(calls_k),
                               "total_ms": round(ms_k, 2), "algorithmic_bytes_per_launch": bytes_k,
                               "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                               # (profiles older than r2e carry the low-precision kernel without its last template argument)
                               "traffic": pmc_traffic(kname_k[:kname_k.rindex(",")] if kname_k.startswith("k_spmv_lp") else kname_k)
                               if cfg == 5 else None}
    roofline = None
    if per_kernel:
        dom = max(per_kernel, key=lambda k_: per_kernel[k_]["total_ms"])       # dominant = largest total time, live
        d = per_kernel[dom]
        fam_bytes = sum(v["algorithmic_bytes_per_launch"] * v["launches"] for v in per_kernel.values())
        fam_ms = sum(v["total_ms"] for v in per_kernel.values())
        roofline = {"bound": "hbm", "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
                    "traffic": d["traffic"], "kernel": d["kernel"], "what": d["what"],
                    "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                    "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                    "selection": "the fine-level SpMV kernel with the largest total time inside the timed region "
                                 "(HIP events around every launch); the four are within a few % of each other",
                    "fine_level_spmv_kernels": per_kernel,
                    "fine_level_spmv_family": {"share_of_step": round(fam_ms / (ms_per_step * args.steps), 3),
                                               "achieved": round(fam_bytes / (fam_ms * 1e-3) / 1e9, 1),
                                               "frac": round(fam_bytes / (fam_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                    "assembly_kernels": {"avg_ms": round(asm_ms, 4), "algorithmic_bytes": asm_bytes,
                                         "achieved": round(asm_bytes / (asm_ms * 1e-3) / 1e9, 1),
                                         "frac": round(asm_bytes / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "nominal: SURVEY 8d's 2480 B/tet over the kernel time; the scratch-free "
                                                 "kernels never move the 2 KiB/tet element matrix (PMC traffic 0.25-0.37x "
                                                 "of that) -- their honest bound is the fp64-VALU figure below",
                                         "path": "scratch-free (k_fused_offdiag + k_fused_diag)"
                                                 if P.options.assembly_fused else "staged (k_element + gathers)",
                                         # secondary: executed fp64 VALU flops of the scratch-free path,
                                         # counted from the gfx950 ISA: 300 fmac + 82 fma + 244 mul + 97 add per
                                         # block contribution = 1105 flop, 16 contributions per tet
                                         "fp64_vector": ({"executed_flops_per_tet": 17680,
                                                          "achieved_tflops": round(17680.0 * s["n_tets"] / (asm_ms * 1e-3) / 1e12, 2),
                                                          "peak_tflops": 78.6,
                                                          "frac": round(17680.0 * s["n_tets"] / (asm_ms * 1e-3) / 1e12 / 78.6, 4)}
                                                         if P.options.assembly_fused else None)}}
    out = {
        "metric": "M-DOF/s (assembly+solve) per Newton iteration",
        "value": round(value, 3), "unit": "M-DOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "precision_note": ("operator, residuals, Krylov recurrences and reductions in f64; the AMG preconditioner's "
                           "smoother/residual passes read a " + {1: "fp32", 2: "row-scaled fp16"}[fmt] + " copy of the level "
                           "matrices (vectors and arithmetic f64; same Krylov iteration counts); the strict all-f64 "
                           "figure of the same steps is under all_f64_preconditioner") if fmt else "all f64",
        "config": {"workload": f"BASELINE config {cfg}: {desc} = {n_tets_global} tets, "
                               f"{n_dof_global} dofs, Re={Re:g}, Newton iteration (assemble J+F, AMG setup, "
                               f"{args.ksp} rtol 1e-8, bt line search)",
                   "parallelism": (f"element partition x{world} ({'x-slabs' if cfg == 5 else 'RCB'}), "
                                   f"{n_tets_global // world} tets per GPU" if world > 1 else "single GPU"),
                   "scaling_note": "strong: the one mesh split N ways (north_star: >= 6x at 8 GPUs on the 10 M-tet duct); "
                                   "the weak layout is under weak_scaling",
                   "newton_log_fnorm_kspits_reason": [(float(f"{a:.3e}"), b, c) for a, b, c in log],
                   "phase_ms_per_step": {"assemble": round(tm.assemble_ms / args.steps, 3),
                                         "pc_setup": round(tm.pc_setup_ms / args.steps, 3),
                                         "krylov": round(tm.krylov_ms / args.steps, 3)},
                   "krylov_loop_last_solve": {"host_syncs": ctr["host_syncs"], "allreduces": ctr["allreduces"],
                                              "halo_exchanges": ctr["exchanges"], "its": log[-1][1] if log else None},
                   "amg_levels": tm.amg_levels, "stokes_its": sres.its},
        "roofline": roofline,
        "all_f64_preconditioner": all_f64,
        "weak_scaling": None,
        "cpu_baseline": None,
    }
    U_host = U.cpu().numpy() if (rank == 0 and world == 1 and not dist_on and host is not None) else None
    P.close()
    del P, U, w
    torch.cuda.empty_cache()

    # ---- second key for N > 1: the weak layout (every GPU keeps the single-GPU share) --------------------------
    if cfg == 5 and dist_on and not args.no_weak:
        sc = float(world) ** (1.0 / 3.0)
        wcells = tuple(int(round(c * sc)) for c in cells)
        try:                                  # a failure of the second key must not cost the headline line
            Pw, nd_w, nt_w, desc_w, _ = build_problem(5, wcells, length, Re, world, rank, local_rank, opts, True)
            Uw, sw = Pw.stokes_solve()
            if sw.reason > 0:
                ms_w, log_w, _ = timed_newton_steps(Pw, Uw, args.steps, args.warmup, world)
                out["weak_scaling"] = {"value": round(nd_w / (ms_w * 1e-3) / 1e6, 3), "unit": "M-DOF/s",
                                       "ms_per_step": round(ms_w, 3), "scaling": "weak",
                                       "workload": f"{desc_w} = {nt_w} tets, {nd_w} dofs ({nt_w // world} tets per GPU)",
                                       "ksp_its": [b for _, b, _ in log_w], "stokes_its": sw.its}
            else:
                out["weak_scaling"] = {"error": f"Stokes solve reason {sw.reason}"}
            Pw.close()
        except Exception as exc:              # noqa: BLE001 -- reported in the line
            out["weak_scaling"] = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        if U_host is not None and not args.no_cpu_baseline:
            mesh, bcs = host
            mask, g = bcs.flatten()
            out["cpu_baseline"] = cpu_baseline(mesh, mask, g, U_host, Re, maxit=args.cpu_maxit)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
