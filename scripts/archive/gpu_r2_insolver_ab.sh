#!/bin/bash
# in-solver A/B of kernel variants: alternate whole bench runs on ONE box and compare the HIP-event kernel averages of the
# timed region (back-to-back launches of one kernel in a harness overstate such differences).
#   SNS_FP64_STEPPED=1: fp64 y = Ax / y = Ax + dot with the stepped loop instead of the up-front loads
#   SNS_LP_STEPPED=1:   fp16 fine-level Jacobi sweep / residual with the stepped loop
for rep in 1 2 3; do
for v in ${VARIANTS:-production fp64_stepped lp_stepped}; do
  unset SNS_FP64_STEPPED SNS_LP_STEPPED
  [ $v = fp64_stepped ] && export SNS_FP64_STEPPED=1
  [ $v = lp_stepped ] && export SNS_LP_STEPPED=1
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo FAILED; tail -3 gpurun_out/sweep_tmp.err; exit 1; }
  python - $v <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
k=d["roofline"]["fine_level_spmv_kernels"]
print(f"{sys.argv[1]:13s} {d['ms_per_step']:8.2f} ms (krylov {d['config']['phase_ms_per_step']['krylov']:.2f})  ax {k['ax']['avg_launch_ms']:.4f}  ax_dot {k['ax_dot']['avg_launch_ms']:.4f}  jacobi {k['jacobi']['avg_launch_ms']:.4f}  resid {k['b_minus_ax']['avg_launch_ms']:.4f}  its {[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]}", flush=True)
PY
done; done
