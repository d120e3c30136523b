#!/bin/bash
# round 4, call s: first contact of the peer-window transport -- 3 / 4 PROCESSES on the box's one GPU (tests/test_gpu_peer.py)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_peer.py -x -q -s -m gpu > gpurun_out/r4s_peer_tests.log 2>&1
rc=$?
tail -30 gpurun_out/r4s_peer_tests.log
exit $rc
