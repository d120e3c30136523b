#!/bin/bash
# round 4, call b: block smoother + dense level: new GPU tests, then slab share / headline with the option combinations
python scripts/gpu_r4_dense_check.py > gpurun_out/r4b_dense_check.log 2>&1; tail -4 gpurun_out/r4b_dense_check.log
timeout -k 10 900 python -m pytest tests/test_gpu_amg.py -x -q -s > gpurun_out/r4b_amg_tests.log 2>&1; tail -25 gpurun_out/r4b_amg_tests.log
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
for rep in 1 2; do
run "slab default (block+dense)" $SLAB
run "slab block0" $SLAB --opt amg_block_smooth=0
run "slab block0 dense0 (round 3)" $SLAB --opt amg_block_smooth=0 --opt amg_dense_rows=0
done
run "slab bnu 1+2,2,1" $SLAB --opt amg_bnu_l1=2
run "slab bnu 1+3,3,1" $SLAB --opt amg_bnu_l2=3
run "slab bnu 1+2,1,1" $SLAB --opt amg_bnu_l1=2 --opt amg_bnu_l2=1
run "10M default" --steps 6 --warmup 2
run "10M block0" --steps 6 --warmup 2 --opt amg_block_smooth=0
run "10M round 3" --steps 6 --warmup 2 --opt amg_block_smooth=0 --opt amg_dense_rows=0
run "10M bnu 1+2" --steps 6 --warmup 2 --opt amg_bnu_l1=2
run "10M bnu l2=3" --steps 6 --warmup 2 --opt amg_bnu_l2=3
run "10M block fine too" --steps 6 --warmup 2 --opt amg_block_smooth=2
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg3 round 3" --config 3 --steps 8 --warmup 2 --opt amg_block_smooth=0 --opt amg_dense_rows=0
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 round 3" --config 4 --steps 4 --warmup 1 --opt amg_block_smooth=0 --opt amg_dense_rows=0
run "cfg4u default" --config 4u --steps 4 --warmup 1
run "cfg4u round 3" --config 4u --steps 4 --warmup 1 --opt amg_block_smooth=0 --opt amg_dense_rows=0
bash scripts/gpu_r4_slab_profile.sh r4b > gpurun_out/r4b_slab_profile.log 2>&1
tail -48 gpurun_out/r4b_slab_profile.log
