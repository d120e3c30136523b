#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
L='{"halo_windows": 0, "amg_exact_sweeps": 0}'
{
for i in 1 2; do echo "== r5 legacy two-stream"; timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 "$L" 2>/dev/null| head -1; done
echo "== r5 windows"; timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 2>/dev/null | head -1
} > gpurun_out/r5e.log 2>&1
grep -v amdgpu.ids gpurun_out/r5e.log | cut -c1-260
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team or partitioned or halo or fgmres_under or rccl" > gpurun_out/r5e_part_tests.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r5e_part_tests.log | cut -c1-400
timeout -k 10 600 python -m pytest tests/test_gpu_peer.py -x -q -m gpu > gpurun_out/r5e_peer_tests.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r5e_peer_tests.log | cut -c1-400
timeout -k 10 300 python scripts/gpu_r5_strong_rehearsal.py 8,4,2 300,75,75 > gpurun_out/r5e_rehearsal.log 2>&1; grep "^N=" gpurun_out/r5e_rehearsal.log | cut -c1-500; tail -5 gpurun_out/r5e_rehearsal.log | cut -c1-300
