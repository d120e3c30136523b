#!/bin/bash
# round 2: fp16 row-scaled matrix copies in the preconditioner vs fp32 copies (one process per variant)
for o in "amg_f32_matrix=1" "amg_f32_matrix=2" "amg_f32_matrix=2 --opt amg_fine_cycle=1" "amg_f32_matrix=1 --opt amg_fine_cycle=1"; do
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 --opt $o > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err
  python - "$o" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1], d["ms_per_step"], d["config"]["newton_log_fnorm_kspits_reason"], d["config"]["phase_ms_per_step"], "jacobi ms", r["avg_launch_ms"], r["other_fine_spmv"], flush=True)
PY
done
