#!/bin/bash
# round 5, call a: first contact of the window form of the partitioned cycle (halo_windows / amg_exact_sweeps) with the GPU:
# the partitioned tests (team + peer), then the 8-way strong rehearsal without the two-stream choreography
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team or partitioned or halo or fgmres_under or rccl" > gpurun_out/r5a_part_tests.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r5a_part_tests.log | cut -c1-400
timeout -k 10 600 python -m pytest tests/test_gpu_peer.py -x -q -m gpu > gpurun_out/r5a_peer_tests.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r5a_peer_tests.log | cut -c1-400
timeout -k 10 300 python scripts/gpu_r5_strong_rehearsal.py 8 300,75,75 > gpurun_out/r5a_rehearsal.log 2>&1; grep "^N=" gpurun_out/r5a_rehearsal.log | cut -c1-500; tail -5 gpurun_out/r5a_rehearsal.log | cut -c1-300
