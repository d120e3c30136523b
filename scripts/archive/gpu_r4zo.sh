#!/bin/bash
# round 4, call zo: XCD-aware workgroup order in the aggregate-block kernels (k_bsweep / k_bpost / k_bfirst / k_restrict_blk / k_resid_restrict)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_amg.py -x -q -m gpu > gpurun_out/r4zo_amg_tests.log 2>&1 || { tail -30 gpurun_out/r4zo_amg_tests.log; exit 1; }
tail -1 gpurun_out/r4zo_amg_tests.log
for rep in 1 2; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f64-rerun > gpurun_out/r4zo_bench_$rep.json 2> gpurun_out/r4zo_bench.err
python - gpurun_out/r4zo_bench_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"10M {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} frac {d['roofline']['frac']}", flush=True)
PY
done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4zo -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f64-rerun > $R/gpurun_out/r4zo_under_rocprof.json 2> $R/gpurun_out/r4zo_stats.err
cd $R
cp $(find gpurun_out/prof_r4zo -name "*kernel_stats.csv" | head -1) gpurun_out/r4zo_bench_kernel_stats.csv; rm -rf gpurun_out/prof_r4zo
python scripts/prof_top.py gpurun_out/r4zo_bench_kernel_stats.csv 14
