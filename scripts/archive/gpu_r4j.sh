#!/bin/bash
# round 4, call j: the record -- full GPU suite (all tests), the unstructured comparison of round 3 re-measured, slab + headline profiles
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4j_gputests.log 2>&1; tail -4 gpurun_out/r4j_gputests.log | cut -c1-200
timeout -k 10 600 python scripts/gpu_r4_unstructured.py 28 > gpurun_out/r4j_unstructured.log 2>&1; tail -5 gpurun_out/r4j_unstructured.log | cut -c1-200
timeout -k 10 600 python scripts/gpu_r4_unstructured.py 28 amg_block_smooth=0 amg_dense_rows=0 amg_ritz_limit=0 > gpurun_out/r4j_unstructured_r3opts.log 2>&1; tail -5 gpurun_out/r4j_unstructured_r3opts.log | cut -c1-200
bash scripts/gpu_r4_slab_profile.sh r4j > gpurun_out/r4j_slab_profile.log 2>&1; head -3 gpurun_out/r4j_slab_profile.log | cut -c1-200
timeout -k 10 900 bash scripts/gpu_profile_round4.sh r4j > gpurun_out/r4j_profile.log 2>&1; tail -3 gpurun_out/r4j_profile.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r4j/r4j_bench_unprofiled.json').read().strip().split("\n")[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['phase_ms_per_step'], [b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']])
PY
