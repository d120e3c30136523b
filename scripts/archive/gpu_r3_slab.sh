#!/bin/bash
# round 3: the 8-way slab share (38x75x75 cells = 1.28 M tets) on one GPU, alternating option sets
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 8 --warmup 2 --cells 38,75,75 --length 0.5 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:32s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']}", flush=True)
PY
}
for rep in 1 2 3; do
run "slab default"
run "slab unfused" --opt amg_fused_post=0
run "slab l2=4 deep=1" --opt amg_nu_l2=4 --opt amg_nu_deep=1
run "slab max_levels 4" --opt amg_max_levels=4
done
