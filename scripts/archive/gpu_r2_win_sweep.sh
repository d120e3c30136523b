#!/bin/bash
# round 2: windowed LDS gathers on/off x fp32 / fp16 preconditioner matrices (one process per variant)
for env in "" "SNS_NO_WINDOWS=1"; do
for o in "amg_f32_matrix=1" "amg_f32_matrix=2"; do
  env $env python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 --opt $o > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err
  python - "$env $o" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], "jacobi ms", r["avg_launch_ms"], {k:v["avg_ms"] for k,v in r["other_fine_spmv"].items()}, flush=True)
PY
done
done
