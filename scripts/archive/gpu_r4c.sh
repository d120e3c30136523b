#!/bin/bash
# round 4, call c
timeout -k 10 600 python -m pytest tests/test_gpu_amg.py -x -q -s > gpurun_out/r4c_amg_tests.log 2>&1; tail -12 gpurun_out/r4c_amg_tests.log
timeout -k 10 120 python scripts/gpu_r4_rccl_floor.py > gpurun_out/r4c_rccl_floor.log 2>&1; tail -2 gpurun_out/r4c_rccl_floor.log
timeout -k 10 600 python scripts/gpu_r4_hardcase.py > gpurun_out/r4c_hardcase.log 2>&1; cat gpurun_out/r4c_hardcase.log | tail -12
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
run "slab default" $SLAB
run "slab round 3" $SLAB --opt amg_block_smooth=0 --opt amg_dense_rows=0
done
run "slab bnu_l2=2" $SLAB --opt amg_bnu_l2=2
run "slab bnu_l1=2" $SLAB --opt amg_bnu_l1=2
run "10M default" --steps 6 --warmup 2
run "10M round 3" --steps 6 --warmup 2 --opt amg_block_smooth=0 --opt amg_dense_rows=0
run "10M block max rows 0 (L1 block too)" --steps 6 --warmup 2 --opt amg_block_max_rows=0
run "10M bnu_l2=2" --steps 6 --warmup 2 --opt amg_bnu_l2=2
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4u default" --config 4u --steps 4 --warmup 1
bash scripts/gpu_r4_slab_profile.sh r4c > gpurun_out/r4c_slab_profile.log 2>&1
tail -44 gpurun_out/r4c_slab_profile.log | cut -c1-130
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path > gpurun_out/r4c_gputests.log 2>&1; tail -15 gpurun_out/r4c_gputests.log
