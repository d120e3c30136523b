#!/bin/bash
# round 4, call m: final GPU suite + weak-layout iteration counts at N = 8 under a few smoother policies (team rehearsal)
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r4m_gputests.log 2>&1; tail -4 gpurun_out/r4m_gputests.log | cut -c1-200
for o in "amg_block_max_rows=8192" "amg_block_max_rows=32768" "amg_block_max_rows=32768,amg_bnu_l2=4" "amg_block_smooth=0"; do
  echo "== weak N=8, $o"; timeout -k 10 300 python scripts/gpu_weak_rehearsal.py 8 300,75,75 $o 2>&1 | grep -E "^N=|owned nodes" | head -2 | cut -c1-220
done
for o in "amg_block_max_rows=8192" "amg_block_max_rows=32768" "amg_block_smooth=0"; do
  echo "== strong N=8, $o"; timeout -k 10 300 python scripts/gpu_r4_strong_rehearsal.py 8 300,75,75 $o 2>&1 | grep -E "^N=" | cut -c1-420
done
