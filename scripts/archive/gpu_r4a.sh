#!/bin/bash
# round 4, call a: dense GJ inverse check + its effect at the slab share and on the headline
set -e
python scripts/gpu_r4_dense_check.py > gpurun_out/r4a_dense_check.log 2>&1 || { tail -20 gpurun_out/r4a_dense_check.log; exit 1; }
cat gpurun_out/r4a_dense_check.log
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
for rep in 1 2; do
run "slab dense512" $SLAB
run "slab dense0" $SLAB --opt amg_dense_rows=0
done
run "10M dense512" --steps 6 --warmup 2
run "10M dense0" --steps 6 --warmup 2 --opt amg_dense_rows=0
run "cfg3 dense512" --config 3 --steps 8 --warmup 2
run "cfg3 dense0" --config 3 --steps 8 --warmup 2 --opt amg_dense_rows=0
run "cfg4u dense512" --config 4u --steps 4 --warmup 1
run "cfg4u dense0" --config 4u --steps 4 --warmup 1 --opt amg_dense_rows=0
bash scripts/gpu_r4_slab_profile.sh r4a_dense > gpurun_out/r4a_slab_profile.log 2>&1
tail -45 gpurun_out/r4a_slab_profile.log
