#!/bin/bash
# round 5, call r5w: the aggregation chooser + the square-lattice nozzle cross-section -- whole suite, benches, rehearsal
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5w_gputests.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r5w_gputests.log | cut -c1-400
for c in 5 4 4u 4b; do
  SNS_AGGREGATION_VERBOSE=1 python bench.py --config $c --no-cpu-baseline --no-f64-rerun > gpurun_out/r5w_bench_$c.json 2> gpurun_out/r5w_bench_$c.err
  python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/r5w_bench_{c}.json") if l.startswith("{")][0])
print(c, d["value"], d["ms_per_step"], [x[1] for x in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], d["config"].get("amg_levels"), d["config"].get("stokes_its"))
PY
  grep "\[sns\]" gpurun_out/r5w_bench_$c.err | head -4
done
python scripts/gpu_r5_strong_rehearsal.py 4,8 300,75,75 2>&1 | grep "^N=" | cut -c1-200
