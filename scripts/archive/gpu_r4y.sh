#!/bin/bash
# round 4, call y: full GPU suite with the peer transport and the rows-per-rank policy for fine-level blocks in the library, then the strong
# layout as 8 threads (team) and as 4 processes (peer windows) with the new defaults
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4y_gputests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r4y_gputests.log
timeout -k 10 500 python scripts/gpu_r4_strong_rehearsal.py 2,4,8 > gpurun_out/r4y_team_strong.log 2>&1; grep "^N=" gpurun_out/r4y_team_strong.log
timeout -k 10 600 python scripts/gpu_r4_peer_strong.py 4 > gpurun_out/r4y_peer_strong.log 2>&1; grep -E "^N=|^      same" gpurun_out/r4y_peer_strong.log
