#!/bin/bash
# round 4, call h: two-stream elimination check, full suite, round profile, team rehearsal
timeout -k 10 300 python scripts/gpu_r4_dense_check.py > gpurun_out/r4h_dense_check.log 2>&1; tail -5 gpurun_out/r4h_dense_check.log
SNS_GJ_ONE_STREAM=1 timeout -k 10 300 python scripts/gpu_r4_dense_check.py > gpurun_out/r4h_dense_check_one_stream.log 2>&1; tail -3 gpurun_out/r4h_dense_check_one_stream.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path > gpurun_out/r4h_gputests.log 2>&1; tail -6 gpurun_out/r4h_gputests.log | cut -c1-220
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
for rep in 1 2 3; do
run "10M default" $T
run "10M round 3 options" $T $R3
run "slab default" $SLAB
run "slab round 3 options" $SLAB $R3
done
timeout -k 10 600 python scripts/gpu_r4_strong_rehearsal.py 1,2,4,8 > gpurun_out/r4h_strong_rehearsal.log 2>&1; tail -6 gpurun_out/r4h_strong_rehearsal.log | cut -c1-420
bash scripts/gpu_r4_slab_profile.sh r4h > gpurun_out/r4h_slab_profile.log 2>&1; tail -42 gpurun_out/r4h_slab_profile.log | cut -c1-120
timeout -k 10 900 bash scripts/gpu_profile_round4.sh r4h > gpurun_out/r4h_profile.log 2>&1; tail -12 gpurun_out/r4h_profile.log
