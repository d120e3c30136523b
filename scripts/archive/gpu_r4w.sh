#!/bin/bash
# round 4, call w: aggregate blocks on the FINE level too (amg_block_smooth = 2, fp16 inverse blocks since call r4o) -- one GPU at 10 M and
# 81 M tets, the slab share, and the 8-way strong split as threads (team) where the fine level's blocks now also run on partitioned handles
mkdir -p gpurun_out
run() {
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
L=gpurun_out/r4w.log
{
T="--steps 6 --warmup 2"
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
F="--opt amg_block_smooth=2"
run "10M default" $T
run "10M fine blocks" $T $F
run "slab default" $SLAB
run "slab fine blocks" $SLAB $F
run "slab default" $SLAB
run "slab fine blocks" $SLAB $F
run "24M fine blocks" --cells 400,100,100 --steps 3 --warmup 1 $F
run "81M fine blocks" --cells 600,150,150 --steps 3 --warmup 1 $F
timeout -k 10 500 python scripts/gpu_r4_strong_rehearsal.py 8 2>&1 | grep "^N="
timeout -k 10 500 python scripts/gpu_r4_strong_rehearsal.py 8 300,75,75 amg_block_smooth=2 2>&1 | grep "^N="
} > $L 2>&1
cat $L
