#!/bin/bash
# round 4, call za: sweep counts of the partitioned hierarchy's latency-bound levels at the 8-way strong split (team rehearsal: iterations)
mkdir -p gpurun_out
L=gpurun_out/r4za.log
{
for o in "amg_bnu_l2=4" "amg_bnu_l2=3" "amg_bnu_l2=2" "amg_bnu_l2=3 amg_bnu_deep=1" "amg_bnu_l2=2 amg_bnu_deep=1" "amg_replicate_rows=400000"; do
  echo "== $o"
  timeout -k 10 300 python scripts/gpu_r4_strong_rehearsal.py 8 300,75,75 $o 2>&1 | grep "^N="
done
} > $L 2>&1
cat $L | cut -c1-420
