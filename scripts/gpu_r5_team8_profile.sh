#!/bin/bash
# round 5: kernel statistics of the 8-way STRONG split of the 10 M-tet duct run as 8 threads on one GPU (team transport = the peer
# transport's kernels on in-process windows, host barriers for the flags; every rank's launches go to the one null stream, so no two
# kernels overlap and the durations are solo durations): kernel time, launches and copies per rank and BiCGStab iteration
# usage (GPU box, repo root): bash scripts/gpu_r5_team8_profile.sh <tag> [N [KEY=VALUE ...]]
set -e
tag=$1; N=${2:-8}; shift; shift || true
R=$(pwd)
out=$R/gpurun_out/team_$tag
mkdir -p $out
python scripts/gpu_r5_strong_rehearsal.py $N 300,75,75 "$@" > $out/unprofiled.log 2>&1
grep "^N=" $out/unprofiled.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/stats -o team -- python3 $R/scripts/gpu_r5_strong_rehearsal.py $N 300,75,75 "$@" > $out/under_rocprof.log 2> $out/stats.err
cd $R
grep "^N=" $out/under_rocprof.log | cut -c1-200
ls $out/stats/* | head
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_team${N}_kernel_stats.csv
cp $(find $out/stats -name "*memory_copy_stats.csv" | head -1) $out/${tag}_team${N}_memory_copy_stats.csv 2>/dev/null || true
rm -rf $out/stats
its=$(grep "^N=" $out/under_rocprof.log | python -c "
import re,sys
l=sys.stdin.read()
m=re.search(r'stokes its (\d+) newton ksp its (\d+),(\d+)',l)
print($N*(int(m.group(1))+int(m.group(2))+int(m.group(3))))")
echo "rank-iterations $its"
python scripts/prof_rank_iteration.py $out/${tag}_team${N}_kernel_stats.csv $its
cat $out/${tag}_team${N}_memory_copy_stats.csv 2>/dev/null | head -8
