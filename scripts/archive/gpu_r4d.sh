#!/bin/bash
# round 4, call d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/gj_tile_bench scripts/r4_micro/gj_tile_bench.hip > gpurun_out/r4d_gj_tile.log 2>&1 && timeout -k 5 60 /tmp/gj_tile_bench >> gpurun_out/r4d_gj_tile.log 2>&1; tail -7 gpurun_out/r4d_gj_tile.log
timeout -k 10 600 python -m pytest tests/test_gpu_amg.py -x -q -s > gpurun_out/r4d_amg_tests.log 2>&1; tail -8 gpurun_out/r4d_amg_tests.log | cut -c1-250
timeout -k 10 900 python scripts/gpu_r4_strong_rehearsal.py 1,2,8 > gpurun_out/r4d_strong_rehearsal.log 2>&1; tail -6 gpurun_out/r4d_strong_rehearsal.log | cut -c1-420
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
T="--steps 6 --warmup 2"
run "10M default" $T
run "10M round 3" $T --opt amg_block_smooth=0 --opt amg_dense_rows=0
run "10M bnu_deep=2" $T --opt amg_bnu_deep=2
run "10M bnu_l2=4" $T --opt amg_bnu_l2=4
run "10M dense 128" $T --opt amg_dense_rows=128
run "10M dense 128 bnu_deep=2" $T --opt amg_dense_rows=128 --opt amg_bnu_deep=2
run "10M block0 dense 512" $T --opt amg_block_smooth=0
run "10M block0 dense 128" $T --opt amg_block_smooth=0 --opt amg_dense_rows=128
run "10M default again" $T
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
run "slab default" $SLAB
run "slab dense 128" $SLAB --opt amg_dense_rows=128
run "slab dense 128 bnu_l2=2" $SLAB --opt amg_dense_rows=128 --opt amg_bnu_l2=2
run "slab round 3" $SLAB --opt amg_block_smooth=0 --opt amg_dense_rows=0
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path > gpurun_out/r4d_gputests.log 2>&1; tail -30 gpurun_out/r4d_gputests.log | cut -c1-220
