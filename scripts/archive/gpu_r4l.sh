#!/bin/bash
# round 4, call l: final record of the test-suite with the final defaults + regression checks outside the benches
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4l_gputests.log 2>&1; tail -4 gpurun_out/r4l_gputests.log | cut -c1-200
timeout -k 10 600 python scripts/gpu_r2_reference_runs.py > gpurun_out/r4l_reference_runs.log 2>&1; grep "^==" gpurun_out/r4l_reference_runs.log | cut -c1-200
timeout -k 10 900 python scripts/gpu_weak_rehearsal.py 2,8 > gpurun_out/r4l_weak_rehearsal.log 2>&1; tail -4 gpurun_out/r4l_weak_rehearsal.log | cut -c1-300
timeout -k 10 300 python scripts/gpu_dfg3d.py > gpurun_out/r4l_dfg3d.log 2>&1; tail -4 gpurun_out/r4l_dfg3d.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4l_smoke.log 2>&1; tail -3 gpurun_out/r4l_smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --gpus 1 > gpurun_out/r4l_bench_driver_command.json 2> gpurun_out/r4l_bench_driver_command.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4l_bench_driver_command.json').read().strip().split("\n")[-1])
print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}, d['config']['phase_ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['cpu_baseline']['ksp_reason'], d['cpu_baseline']['value'])
PY
