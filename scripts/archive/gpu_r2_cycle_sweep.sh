#!/bin/bash
# round 2: fine-level cycle shape and coarse sweep counts on the headline workload (one process per variant)
for o in "amg_fine_cycle=0" "amg_fine_cycle=1" "amg_fine_cycle=2" "amg_nu_coarse=2" "amg_nu_coarse=3" "amg_nu_l2=4" "amg_nu_coarse=3 --opt amg_nu_l2=4"; do
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 3 --opt $o > gpurun_out/sweep_tmp.json 2>/dev/null
  python - "$o" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], flush=True)
PY
done
