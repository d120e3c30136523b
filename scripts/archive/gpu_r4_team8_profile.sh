#!/bin/bash
# round 4: kernel statistics of the 8-way STRONG split of the 10 M-tet duct run as 8 threads on one GPU (team transport; every rank's
# launches go to the one null stream, so no two kernels overlap and the durations are solo durations): the N-independent kernel time
# per rank and BiCGStab iteration of the PARTITIONED code path (one level more than the single-GPU slab share, split level-0 passes)
# usage (GPU box, repo root): bash scripts/gpu_r4_team8_profile.sh <tag> [N [KEY=VALUE ...]]
set -e
tag=$1; N=${2:-8}; shift; shift || true
R=$(pwd)
out=$R/gpurun_out/team_$tag
mkdir -p $out
python scripts/gpu_r4_strong_rehearsal.py $N 300,75,75 "$@" > $out/unprofiled.log 2>&1
grep "^N=" $out/unprofiled.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o team -- python3 $R/scripts/gpu_r4_strong_rehearsal.py $N 300,75,75 "$@" > $out/under_rocprof.log 2> $out/stats.err
cd $R
grep "^N=" $out/under_rocprof.log
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_team${N}_kernel_stats.csv
rm -rf $out/stats
python scripts/prof_top.py $out/${tag}_team${N}_kernel_stats.csv 45
