#!/bin/bash
# round 3: does more coarse-level smoothing buy iterations at 81 M tets (8 levels)?
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 2 --warmup 0 --cells 600,150,150 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:40s} {d['ms_per_step']:9.1f} ms/step its {its} stokes {d['config']['stokes_its']} levels {d['config']['amg_levels']}", flush=True)
PY
}
run "81M default"
run "81M deep 4" --opt amg_nu_deep=4
run "81M l2 8 deep 4" --opt amg_nu_l2=8 --opt amg_nu_deep=4
run "81M coarse 6 (L1 1+8)" --opt amg_nu_coarse=6
run "81M L1 4+4" --opt amg_nu_l1_pre=4 --opt amg_nu_l1_post=4
run "81M fine nu 2" --opt amg_nu=2
