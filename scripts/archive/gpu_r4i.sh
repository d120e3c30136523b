#!/bin/bash
# round 4, call i: interleaved A/B on one box (5 rounds), large meshes, final profile
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
R3="--opt amg_block_smooth=0 --opt amg_dense_rows=0 --opt amg_ritz_limit=0"
for rep in 1 2 3 4 5; do
run "10M default" $T
run "10M round 3 options" $T $R3
run "slab default" $SLAB
run "slab round 3 options" $SLAB $R3
done
run "24M default" --cells 400,100,100 --steps 3 --warmup 1
run "24M round 3 options" --cells 400,100,100 --steps 3 --warmup 1 $R3
run "81M default" --cells 600,150,150 --steps 3 --warmup 1
run "81M round 3 options" --cells 600,150,150 --steps 3 --warmup 1 $R3
