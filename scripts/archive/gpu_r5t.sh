#!/bin/bash
# round 5, call r5t: 1 + 4 exact sweeps on the partitioned level 1 -- partitioned tests, rehearsal at N = 2, 4, 8, team8 profile
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer.py tests/test_gpu_amg.py -x -q -m gpu -k "team or partitioned or halo or fgmres_under or rccl or peer or window_cycle or entry_point or bench" > gpurun_out/r5t_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r5t_tests.log | cut -c1-400
python scripts/gpu_r5_strong_rehearsal.py 2,4 300,75,75 2>&1 | grep "^N=" | cut -c1-330
bash scripts/gpu_r5_team8_profile.sh r5t 8 > gpurun_out/r5t_team8.log 2>&1; grep -A12 "rank-iterations" gpurun_out/r5t_team8.log | cut -c1-200; grep "^N=" gpurun_out/team_r5t/unprofiled.log | cut -c1-330
