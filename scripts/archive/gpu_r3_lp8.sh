#!/bin/bash
# round 3: fp16 preconditioner passes with 8 lanes per row (full 128-B lines per load instruction), A/B in the solver
run() {
  env $1 python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
k=d["roofline"]["fine_level_spmv_kernels"]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]} jacobi {k['jacobi']['avg_launch_ms']*1e3:6.1f} us ({k['jacobi']['frac']:.3f}) resid {k['b_minus_ax']['avg_launch_ms']*1e3:6.1f} us ({k['b_minus_ax']['frac']:.3f}) ax {k['ax']['avg_launch_ms']*1e3:6.1f}", flush=True)
PY
}
SNS_LP8=1 SNS_LP8_COARSE=7 python -m pytest tests/test_gpu_parity.py -x -q -k "low_precision or block_jacobi or stokes_solve_vs_lu or team_transport or delaunay or two_stream" 2>&1 | tail -3
SNS_LP8=2 SNS_LP8_COARSE=4 python -m pytest tests/test_gpu_parity.py -x -q -k "low_precision or block_jacobi or stokes_solve_vs_lu or delaunay" 2>&1 | tail -3
for rep in 1 2; do
run "SNS_LP8=0"
run "SNS_LP8=1"
run "SNS_LP8=2"
run "SNS_LP8=1 SNS_LP8_COARSE=4"
run "SNS_LP8=1 SNS_LP8_COARSE=7"
run "SNS_LP8=2 SNS_LP8_COARSE=7"
done
