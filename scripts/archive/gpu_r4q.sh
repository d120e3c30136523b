#!/bin/bash
# round 4, call q: sweep counts of the aggregate-block levels around the new defaults (1 + 3, 4 + 4, 2 + 2)
run() {
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
T="--steps 8 --warmup 2"
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
run "10M default 1+3 / 4 / 2" $T
run "10M l1 post 4" $T --opt amg_bnu_l1=4
run "10M l1 post 2" $T --opt amg_bnu_l1=2
run "10M l2 = 5" $T --opt amg_bnu_l2=5
run "10M deep = 3" $T --opt amg_bnu_deep=3
run "10M deep = 1" $T --opt amg_bnu_deep=1
run "10M l1 pre 2 post 3" $T --opt amg_nu_l1_pre=2 --opt amg_nu_l1_post=3
run "10M default again" $T
run "slab default" $SLAB
run "slab l2 = 3" $SLAB --opt amg_bnu_l2=3
run "slab l1 post 2" $SLAB --opt amg_bnu_l1=2
run "slab deep..." $SLAB --opt amg_bnu_l1=4
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 l1 post 4" --config 4 --steps 4 --warmup 1 --opt amg_bnu_l1=4
run "cfg4u default" --config 4u --steps 4 --warmup 1
run "cfg4u l1 post 4" --config 4u --steps 4 --warmup 1 --opt amg_bnu_l1=4
run "cfg4u l2 = 5" --config 4u --steps 4 --warmup 1 --opt amg_bnu_l2=5
