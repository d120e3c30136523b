#!/bin/bash
# round 4, call x: where do aggregate blocks on the fine level pay?  configs 3 / 4 / 4u on one GPU, the 1/4 and 1/2 slab shares, team N = 2 / 4
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
L=gpurun_out/r4x.log
{
F="--opt amg_block_smooth=2"
for rep in 1 2; do
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg3 fine blocks" --config 3 --steps 8 --warmup 2 $F
done
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 fine blocks" --config 4 --steps 4 --warmup 1 $F
run "cfg4u default" --config 4u --steps 4 --warmup 1
run "cfg4u fine blocks" --config 4u --steps 4 --warmup 1 $F
run "quarter slab default" --steps 6 --warmup 2 --cells 75,75,75 --length 1.0
run "quarter slab fine blocks" --steps 6 --warmup 2 --cells 75,75,75 --length 1.0 $F
run "half slab default" --steps 6 --warmup 2 --cells 150,75,75 --length 2.0
run "half slab fine blocks" --steps 6 --warmup 2 --cells 150,75,75 --length 2.0 $F
timeout -k 10 500 python scripts/gpu_r4_strong_rehearsal.py 2,4 2>&1 | grep "^N="
timeout -k 10 500 python scripts/gpu_r4_strong_rehearsal.py 2,4 300,75,75 amg_block_smooth=2 2>&1 | grep "^N="
} > $L 2>&1
cat $L
