#!/bin/bash
# round 4, call zk: residual + restriction + next first sweep as one launch below the fine level (k_resid_restrict, amg_fuse_restrict) --
# the cycle against the scipy restatement first, the full suite, A/B on the 10 M-tet duct and the slab share, the 8-way team profile
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_amg.py -x -q -m gpu > gpurun_out/r4zk_amg_tests.log 2>&1 || { tail -30 gpurun_out/r4zk_amg_tests.log; exit 1; }
tail -2 gpurun_out/r4zk_amg_tests.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4zk_gputests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4zk_gputests.log | cut -c1-300
run() {
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
{
T="--steps 8 --warmup 2"
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
for rep in 1 2; do
run "10M fused restriction" $T
run "10M separate kernels" $T --opt amg_fuse_restrict=0
run "slab fused restriction" $SLAB
run "slab separate kernels" $SLAB --opt amg_fuse_restrict=0
done
run "cfg3 fused" --config 3 --steps 8 --warmup 2
run "cfg3 separate" --config 3 --steps 8 --warmup 2 --opt amg_fuse_restrict=0
} > gpurun_out/r4zk.log 2>&1
cat gpurun_out/r4zk.log
bash scripts/gpu_r4_team8_profile.sh r4zk 8 > gpurun_out/r4zk_team8.log 2>&1; grep "^N=" gpurun_out/r4zk_team8.log | cut -c1-200
python scripts/prof_rank_iteration.py gpurun_out/team_r4zk/r4zk_team8_kernel_stats.csv 776 | head -8
