#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python scripts/gpu_r5_nozzle_variants.py 2>&1 | grep -v amdgpu.ids | grep "body-fitted lc\|staircase (" > gpurun_out/r5m_nozzle.log; cat gpurun_out/r5m_nozzle.log | cut -c1-250
for c in 4b 5; do
timeout -k 10 600 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-f64-rerun > gpurun_out/r5m_bench_$c.json 2> gpurun_out/r5m_bench_$c.err; echo "bench $c rc $?"
python - $c <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r5m_bench_{sys.argv[1]}.json").read().strip().split("\n")[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"])
PY
done
