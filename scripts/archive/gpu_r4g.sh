#!/bin/bash
# round 4, call g: full GPU suite, A/B against round 3's options, slab profile
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path > gpurun_out/r4g_gputests.log 2>&1; tail -6 gpurun_out/r4g_gputests.log | cut -c1-220
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
run "10M round 3 options" $T $R3
run "10M block off" $T --opt amg_block_smooth=0
run "10M bnu_deep=1" $T --opt amg_bnu_deep=1
done
for rep in 1 2; do
run "slab default" $SLAB
run "slab round 3 options" $SLAB $R3
run "slab block off" $SLAB --opt amg_block_smooth=0
run "slab bnu_l2=2" $SLAB --opt amg_bnu_l2=2
done
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg3 round 3 options" --config 3 --steps 8 --warmup 2 $R3
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 round 3 options" --config 4 --steps 4 --warmup 1 $R3
run "cfg4u default" --config 4u --steps 4 --warmup 1
run "cfg4u round 3 options" --config 4u --steps 4 --warmup 1 $R3
bash scripts/gpu_r4_slab_profile.sh r4g > gpurun_out/r4g_slab_profile.log 2>&1
tail -46 gpurun_out/r4g_slab_profile.log | cut -c1-130
