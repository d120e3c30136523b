#!/bin/bash
# round 2: fewer sweeps on the deep levels (amg_nu_l2 6 -> 4, amg_nu_deep 2 -> 1) at 10 M tets and at the 8-way slab size
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(f"{sys.argv[1]:40s} {d['ms_per_step']:8.2f} ms  its {[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]}", flush=True)
PY
}
for rep in 1 2; do
run "10M default (1,4,6,2)"
run "10M l2=4 deep=1" --opt amg_nu_l2=4 --opt amg_nu_deep=1
run "10M l2=4 deep=2" --opt amg_nu_l2=4
run "10M l2=5 deep=1" --opt amg_nu_l2=5 --opt amg_nu_deep=1
done
for rep in 1 2; do
run "slab default" --cells 38,75,75 --length 0.5
run "slab l2=4 deep=1" --cells 38,75,75 --length 0.5 --opt amg_nu_l2=4 --opt amg_nu_deep=1
run "slab l2=4 deep=2" --cells 38,75,75 --length 0.5 --opt amg_nu_l2=4
run "slab coarse=3 l2=4 deep=1" --cells 38,75,75 --length 0.5 --opt amg_nu_coarse=3 --opt amg_nu_l2=4 --opt amg_nu_deep=1
done
