#!/bin/bash
# round 4, call f
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/gj_tile_bench scripts/r4_micro/gj_tile_bench.hip > gpurun_out/r4f_gj_tile.log 2>&1 && timeout -k 5 60 /tmp/gj_tile_bench >> gpurun_out/r4f_gj_tile.log 2>&1; tail -9 gpurun_out/r4f_gj_tile.log
timeout -k 10 300 python scripts/gpu_r4_dense_check.py > gpurun_out/r4f_dense_check.log 2>&1; tail -4 gpurun_out/r4f_dense_check.log
timeout -k 10 600 python -m pytest tests/test_gpu_amg.py -x -q -s > gpurun_out/r4f_amg_tests.log 2>&1; tail -4 gpurun_out/r4f_amg_tests.log | cut -c1-250
timeout -k 10 600 python scripts/gpu_r4_hardcase.py > gpurun_out/r4f_hardcase.log 2>&1; cat gpurun_out/r4f_hardcase.log | tail -9 | cut -c1-330
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
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
for rep in 1 2; do
run "10M default" $T
run "10M dense 128" $T --opt amg_dense_rows=128
run "10M dense 0" $T --opt amg_dense_rows=0
run "10M round 3 options" $T $R3
done
for rep in 1 2; do
run "slab default" $SLAB
run "slab dense 128" $SLAB --opt amg_dense_rows=128
run "slab dense 0" $SLAB --opt amg_dense_rows=0
run "slab round 3 options" $SLAB $R3
done
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg3 dense 128" --config 3 --steps 8 --warmup 2 --opt amg_dense_rows=128
run "cfg3 round 3 options" --config 3 --steps 8 --warmup 2 $R3
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 dense 128" --config 4 --steps 4 --warmup 1 --opt amg_dense_rows=128
run "cfg4 round 3 options" --config 4 --steps 4 --warmup 1 $R3
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path > gpurun_out/r4f_gputests.log 2>&1; tail -8 gpurun_out/r4f_gputests.log | cut -c1-220
