#!/bin/bash
# round 4: per-kernel breakdown of the 8-way slab share (38x75x75 cells = 1.28 M tets) on one GPU, plus the un-profiled figure
# usage (GPU box, repo root): bash scripts/gpu_r4_slab_profile.sh <tag> [extra bench args]
set -e
tag=$1; shift
R=$(pwd)
out=$R/gpurun_out/slab_$tag
mkdir -p $out
for rep in 1 2 3; do
python bench.py --no-cpu-baseline --no-f64-rerun --steps 8 --warmup 2 --cells 38,75,75 --length 0.5 "$@" > $out/unprofiled_$rep.json 2> $out/unprofiled.err
python - $out/unprofiled_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"slab {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']}", flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-f64-rerun --cells 38,75,75 --length 0.5 "$@" > $out/under_rocprof.json 2> $out/stats.err
cd $R
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_slab_kernel_stats.csv
rm -rf $out/stats
python scripts/prof_top.py $out/${tag}_slab_kernel_stats.csv 40
