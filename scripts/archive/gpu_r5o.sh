#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5o_gputests.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r5o_gputests.log | cut -c1-300
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f64-rerun > gpurun_out/r5o_bench.json 2> gpurun_out/r5o_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5o_bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"])
PY
