#!/bin/bash
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 $1 > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], flush=True)
PY
}
run ""
run "--opt amg_nu_l2=8"
run "--opt amg_nu_l2=10"
run "--opt amg_nu_l2=12"
run "--opt amg_nu_l2=8 --opt amg_nu_deep=4"
run "--opt amg_nu_l2=8 --opt amg_nu_deep=6"
run "--opt amg_nu_l2=10 --opt amg_nu_deep=4"
run "--opt amg_nu_l2=8 --opt amg_nu_deep=3"
