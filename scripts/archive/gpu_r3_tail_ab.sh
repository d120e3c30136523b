#!/bin/bash
# round 3: in-solver A/B of the fused tail cycle (k_tail_cycle; slower, removed -- profiles/r3_tail_kernel_experiment.txt holds the code; amg_fused_tail no longer exists)
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:32s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']}", flush=True)
PY
}
for rep in 1 2; do
run "slab tail" --steps 8 --warmup 2 --cells 38,75,75 --length 0.5
run "slab launch-per-pass" --steps 8 --warmup 2 --cells 38,75,75 --length 0.5 --opt amg_fused_tail=0
run "headline tail" --steps 4 --warmup 1
run "headline launch-per-pass" --steps 4 --warmup 1 --opt amg_fused_tail=0
run "config 3 tail" --config 3 --steps 4 --warmup 1
run "config 3 launch-per-pass" --config 3 --steps 4 --warmup 1 --opt amg_fused_tail=0
run "config 4 tail" --config 4 --steps 4 --warmup 1
run "config 4 launch-per-pass" --config 4 --steps 4 --warmup 1 --opt amg_fused_tail=0
done
