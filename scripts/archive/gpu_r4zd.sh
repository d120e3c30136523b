#!/bin/bash
# round 4, call zd: the round's final record with the final library -- headline profile (scripts/gpu_profile_round4.sh) and the kernel
# statistics of the peer transport's protocol self-test
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 bash scripts/gpu_profile_round4.sh r4zd > gpurun_out/r4zd_profile.log 2>&1; tail -3 gpurun_out/r4zd_profile.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r4zd/r4zd_bench_unprofiled.json').read().strip().split("\n")[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['phase_ms_per_step'], [b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']], d['roofline']['frac'], d['roofline']['step_frac'], d['all_f64_preconditioner'], d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
PY
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4zd/peer -o peer -- python3 $R/scripts/gpu_r4_peer_selftest.py 2 5776 500 > $R/gpurun_out/prof_r4zd/peer_selftest_under_rocprof.log 2>&1
cd $R
cp $(find gpurun_out/prof_r4zd/peer -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r4zd/r4zd_peer_selftest_kernel_stats.csv && rm -rf gpurun_out/prof_r4zd/peer
tail -2 gpurun_out/prof_r4zd/peer_selftest_under_rocprof.log; python scripts/prof_top.py gpurun_out/prof_r4zd/r4zd_peer_selftest_kernel_stats.csv 10
