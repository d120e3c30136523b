#!/bin/bash
# round 3: fine-level sweep count against the coarse schedule, after the fp16 copies made fine passes cheap
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(f"{sys.argv[1]:44s} {d['ms_per_step']:8.2f} ms  its {[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]} {d['config']['phase_ms_per_step']}", flush=True)
PY
}
run "default (1,4,6,2)"
run "nu=2 (2,4,6,2)" --opt amg_nu=2
run "nu=2, coarse 3" --opt amg_nu=2 --opt amg_nu_coarse=3
run "nu=2, coarse 2, l2 4" --opt amg_nu=2 --opt amg_nu_coarse=2 --opt amg_nu_l2=4
run "nu=2, L1 1+6" --opt amg_nu=2 --opt amg_nu_l1_pre=1 --opt amg_nu_l1_post=6
run "nu=3" --opt amg_nu=3
run "default (1,4,6,2)"
