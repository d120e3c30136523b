#!/bin/bash
# round 2 (after the fp16 preconditioner matrices): coarse sweep counts and the level from which the cycle is one hipGraph
run() {
  env $1 python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 $2 > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err
  python - "$1 $2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(sys.argv[1], d["ms_per_step"], [b for a,b,c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"], flush=True)
PY
}
run "" ""
run "SNS_GRAPH_ROWS=250000" ""
run "" "--opt amg_nu_coarse=5"
run "" "--opt amg_nu_coarse=6"
run "" "--opt amg_nu_l2=8"
run "" "--opt amg_nu_coarse=3"
run "SNS_GRAPH_ROWS=250000" "--opt amg_nu_coarse=5"
