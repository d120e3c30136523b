#!/bin/bash
# round 4, call k: the growth check of the damping limited to levels with >= 3 sweeps (amg_growth_check = 1) against every level (2)
timeout -k 10 600 python scripts/gpu_r4_hardcase.py > gpurun_out/r4k_hardcase.log 2>&1; cat gpurun_out/r4k_hardcase.log | tail -10 | cut -c1-330
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
for rep in 1 2 3; do
run "10M default" $T
run "10M growth check 1" $T --opt amg_growth_check=1
run "10M growth check 0" $T --opt amg_growth_check=0
done
run "slab default" $SLAB
run "slab growth check 1" $SLAB --opt amg_growth_check=1
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg3 growth check 1" --config 3 --steps 8 --warmup 2 --opt amg_growth_check=1
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 growth check 1" --config 4 --steps 4 --warmup 1 --opt amg_growth_check=1
run "cfg4u default" --config 4u --steps 4 --warmup 1
run "cfg4u growth check 1" --config 4u --steps 4 --warmup 1 --opt amg_growth_check=1
run "24M default" --cells 400,100,100 --steps 3 --warmup 1
run "24M growth check 1" --cells 400,100,100 --steps 3 --warmup 1 --opt amg_growth_check=1
timeout -k 10 300 python scripts/gpu_r4_unstructured.py 28 amg_growth_check=1 > gpurun_out/r4k_unstructured_gc1.log 2>&1; tail -4 gpurun_out/r4k_unstructured_gc1.log | cut -c1-200
