#!/bin/bash
# round 4, call r: the final record -- full GPU suite, slab + headline profiles with the final defaults
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4r_gputests.log 2>&1; tail -4 gpurun_out/r4r_gputests.log | cut -c1-200
bash scripts/gpu_r4_slab_profile.sh r4r > gpurun_out/r4r_slab_profile.log 2>&1; head -3 gpurun_out/r4r_slab_profile.log | cut -c1-200
timeout -k 10 900 bash scripts/gpu_profile_round4.sh r4r > gpurun_out/r4r_profile.log 2>&1; tail -3 gpurun_out/r4r_profile.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r4r/r4r_bench_unprofiled.json').read().strip().split("\n")[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['phase_ms_per_step'], [b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']], d['roofline']['frac'], d['roofline']['step_frac'], d['all_f64_preconditioner'])
PY
