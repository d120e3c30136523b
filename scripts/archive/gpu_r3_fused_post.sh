#!/bin/bash
# round 3: fused coarse-grid correction + first post-smoothing sweep (M = A P), in-solver A/B
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(f"{sys.argv[1]:40s} {d['ms_per_step']:8.2f} ms  its {[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]} stokes {d['config']['stokes_its']} {d['config']['phase_ms_per_step']}", flush=True)
PY
}
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_post or block_jacobi or stokes_solve_vs_lu or team_transport or low_precision or two_stream or option_combinations" 2>&1 | tail -5
for rep in 1 2; do
run "fused post (default)"
run "unfused" --opt amg_fused_post=0
run "fused + L1 1+6" --opt amg_nu_l1_pre=1 --opt amg_nu_l1_post=6
run "fused config 3" --config 3
run "unfused config 3" --config 3 --opt amg_fused_post=0
run "fused config 4" --config 4
run "unfused config 4" --config 4 --opt amg_fused_post=0
done
