#!/bin/bash
# round 2: sweep counts at the size of ONE rank's share of the 8-way strong split (38x75x75 cells = 1.28 M tets)
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 --cells 38,75,75 --length 0.5067 $1 > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], flush=True)
PY
}
run ""
run "--opt amg_nu_l2=4"
run "--opt amg_nu_l2=3"
run "--opt amg_nu_l2=4 --opt amg_nu_deep=1"
run "--opt amg_nu_coarse=3 --opt amg_nu_l2=4"
run "--opt amg_nu_coarse=2 --opt amg_nu_l2=3 --opt amg_nu_deep=1"
run "--opt amg_coarse_size=64"
run "--opt amg_nu_l2=4 --opt amg_coarse_size=64"
