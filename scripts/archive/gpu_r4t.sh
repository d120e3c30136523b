#!/bin/bash
# round 4, call t: peer-window transport after the fused BiCGStab reduction -- the process tests, then the strong layout with
# N = 2 and 4 processes on the one GPU (scripts/gpu_r4_peer_strong.py), the team rehearsal of the same split beside it
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_peer.py -x -q -s -m gpu > gpurun_out/r4t_peer_tests.log 2>&1 || { tail -30 gpurun_out/r4t_peer_tests.log; exit 1; }
tail -8 gpurun_out/r4t_peer_tests.log
timeout -k 10 1000 python scripts/gpu_r4_peer_strong.py 2,4 > gpurun_out/r4t_peer_strong.log 2>&1 || { tail -30 gpurun_out/r4t_peer_strong.log; exit 1; }
cat gpurun_out/r4t_peer_strong.log
timeout -k 10 600 python scripts/gpu_r4_strong_rehearsal.py 4 > gpurun_out/r4t_team_strong.log 2>&1 || { tail -30 gpurun_out/r4t_team_strong.log; exit 1; }
cat gpurun_out/r4t_team_strong.log
