#!/bin/bash
# round 4, call zh (and zi, with the copy-free replicated tail): fine-level ghost-tail trims of the partitioned cycle (no clearing, no copy into the exchange vector, the cycle in the caller's
# vector) -- full GPU suite, then the per-rank kernel statistics of the 8-way split again
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4zj_gputests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r4zj_gputests.log | cut -c1-300
bash scripts/gpu_r4_team8_profile.sh r4zj 8 > gpurun_out/r4zj_team8.log 2>&1; grep "^N=" gpurun_out/r4zj_team8.log | cut -c1-330
python scripts/prof_rank_iteration.py gpurun_out/team_r4zj/r4zj_team8_kernel_stats.csv 776
