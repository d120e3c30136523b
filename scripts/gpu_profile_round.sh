#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/gpu_profile_round4.sh r4h
# the round's profile: un-profiled default bench (incl. CPU baseline and the all-fp64 repetition), kernel stats under rocprofv3,
# FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only), trimmed to the fine-level family + assembly
set -e
tag=$1
R=$(pwd)
out=$R/gpurun_out/prof_$tag
mkdir -p $out
python bench.py > $out/${tag}_bench_unprofiled.json 2> $out/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f64-rerun > $out/${tag}_bench_under_rocprof.json 2> $out/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f64-rerun > $out/fetch.json 2> $out/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f64-rerun > $out/write.json 2> $out/write.err
echo "write done"
cd $R
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
python scripts/trim_pmc.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_fetch_counter_collection.csv
python scripts/trim_pmc.py $(find $out/write -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_write_counter_collection.csv
rm -rf $out/stats $out/fetch $out/write
ls -la $out
