#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
L='{"halo_windows": 0, "amg_exact_sweeps": 0}'
{
for i in 1 2 3; do echo "== r4 tree"; SNS_TREE=$PWD/.r4ref timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 | head -1; done
for i in 1 2 3; do echo "== r5 legacy two-stream"; timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 "$L" | head -1; done
for i in 1 2; do echo "== r5 legacy two-stream, no link check"; SNS_TEST_CHECK_ROUNDS=0 timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 "$L" | head -1; done
for i in 1 2; do echo "== r5 windows"; timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 | head -1; done
} > gpurun_out/r5d.log 2>&1
grep -v amdgpu.ids gpurun_out/r5d.log | cut -c1-330
