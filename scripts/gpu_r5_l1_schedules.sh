#!/bin/bash
# round 5: iteration counts of the 4- and 8-way strong split against the schedule of the partitioned level 1 (exact global sweeps:
# pre + post counts; amg_exact_sweeps = 0: round 4's 4 + 4 rank-local sweeps)
mkdir -p gpurun_out
for o in "" "amg_bnu_l1=4" "amg_nu_l1_pre=2 amg_nu_l1_post=3" "amg_nu_l1_pre=2 amg_nu_l1_post=4" "amg_nu_l1_pre=2 amg_nu_l1_post=2" "amg_exact_sweeps=0"; do
  echo "== $o"
  python scripts/gpu_r5_strong_rehearsal.py 4,8 300,75,75 $o 2>&1 | grep "^N=" | cut -c1-210
done
