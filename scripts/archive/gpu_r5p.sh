#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer.py tests/test_gpu_amg.py tests/test_gpu_2d.py -x -q -m gpu -k "team or partitioned or halo or fgmres_under or rccl or peer or bench or window_cycle or bodyfitted or tell_apart or entry_point" > gpurun_out/r5p_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r5p_tests.log | cut -c1-400
bash scripts/gpu_r5_team8_profile.sh r5p 8 > gpurun_out/r5p_team8.log 2>&1; grep -A14 "rank-iterations" gpurun_out/r5p_team8.log | cut -c1-200
