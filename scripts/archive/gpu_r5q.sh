#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer.py tests/test_gpu_amg.py -x -q -m gpu -k "team or partitioned or halo or fgmres_under or rccl or peer or window_cycle or entry_point" > gpurun_out/r5q_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r5q_tests.log | cut -c1-400
bash scripts/gpu_r5_team8_profile.sh r5q 8 > gpurun_out/r5q_team8.log 2>&1; grep -A14 "rank-iterations" gpurun_out/r5q_team8.log | cut -c1-200
SNS_NO_CARRIED_PUT=1 python scripts/gpu_r5_strong_rehearsal.py 8 300,75,75 2>&1 | grep "^N=" | cut -c1-300
