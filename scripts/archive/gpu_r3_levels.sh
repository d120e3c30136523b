#!/bin/bash
# round 3: how deep does the hierarchy have to go?  amg_max_levels at 10 M tets and at the 8-way slab share (38x75x75)
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:40s} {d['ms_per_step']:8.2f} ms  its {its} levels {d['config']['amg_levels']} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']}", flush=True)
PY
}
for ml in 12 6 5 4 3; do run "10M max_levels $ml" --opt amg_max_levels=$ml; done
for ml in 12 5 4 3; do run "slab max_levels $ml" --cells 38,75,75 --length 0.5 --opt amg_max_levels=$ml; done
run "slab l2=4 deep=1" --cells 38,75,75 --length 0.5 --opt amg_nu_l2=4 --opt amg_nu_deep=1
run "slab unfused" --cells 38,75,75 --length 0.5 --opt amg_fused_post=0
