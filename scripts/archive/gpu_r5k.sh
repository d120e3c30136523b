#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "bodyfitted" > gpurun_out/r5k_tests.log 2>&1; echo "pytest rc $?"; grep -v amdgpu.ids gpurun_out/r5k_tests.log | tail -12 | cut -c1-500
for c in 4b 4; do
timeout -k 10 600 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-f64-rerun > gpurun_out/r5k_bench_$c.json 2> gpurun_out/r5k_bench_$c.err; echo "bench $c rc $?"
python - $c <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r5k_bench_{sys.argv[1]}.json").read().strip().split("\n")[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], d["config"]["workload"][:200])
PY
done
