#!/bin/bash
# round 4, call ze: peer tests after the self-test learnt the long forms (40-double all-reduce, chunked all-gather)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_peer.py -x -q -s -m gpu > gpurun_out/r4ze_peer_tests.log 2>&1; echo "pytest rc $?"; grep -v "^  duct\|^  cavity" gpurun_out/r4ze_peer_tests.log | tail -8
