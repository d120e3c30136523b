#!/bin/bash
# round 4, call zp: the round's final record with the final library (after k_resid_restrict and the partitioned-cycle trims): headline profile
mkdir -p gpurun_out
timeout -k 10 1000 bash scripts/gpu_profile_round4.sh r4zp > gpurun_out/r4zp_profile.log 2>&1; tail -2 gpurun_out/r4zp_profile.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r4zp/r4zp_bench_unprofiled.json').read().strip().split("\n")[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['phase_ms_per_step'], [b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']], d['roofline']['frac'], d['roofline']['step_frac'], d['all_f64_preconditioner'], d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
PY
python scripts/prof_top.py gpurun_out/prof_r4zp/r4zp_bench_kernel_stats.csv 24
