#!/bin/bash
# round 3: sweeps on level 2 / the deep levels against the mesh size
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps ${STEPS:-2} --warmup 0 --cells $2 "${@:3}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:40s} {d['ms_per_step']:9.1f} ms/step its {its} stokes {d['config']['stokes_its']} levels {d['config']['amg_levels']}", flush=True)
PY
}
for cells in 300,75,75 400,100,100 600,150,150; do
  run "$cells default (l2 6, deep 2)" $cells
  run "$cells deep 4" $cells --opt amg_nu_deep=4
  run "$cells l2 8 deep 4" $cells --opt amg_nu_l2=8 --opt amg_nu_deep=4
  run "$cells l2 8 deep 6" $cells --opt amg_nu_l2=8 --opt amg_nu_deep=6
  run "$cells l2 10 deep 8" $cells --opt amg_nu_l2=10 --opt amg_nu_deep=8
done
STEPS=4 run "10M 4 steps default" 300,75,75
STEPS=4 run "10M 4 steps l2 8 deep 4" 300,75,75 --opt amg_nu_l2=8 --opt amg_nu_deep=4
STEPS=4 run "10M 4 steps deep 4" 300,75,75 --opt amg_nu_deep=4
