#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "== r4 tree"; SNS_TREE=$PWD/.r4ref timeout -k 10 300 python scripts/gpu_r5_teamcheck.py 2>&1 | grep -v amdgpu.ids
echo "== r5 tree, legacy"; timeout -k 10 300 python scripts/gpu_r5_teamcheck.py halo_windows=0 amg_exact_sweeps=0 2>&1 | grep -v amdgpu.ids
echo "== r5 tree, windows"; timeout -k 10 300 python scripts/gpu_r5_teamcheck.py halo_windows=1 amg_exact_sweeps=0 2>&1 | grep -v amdgpu.ids
echo "== r5 tree, windows + exact"; timeout -k 10 300 python scripts/gpu_r5_teamcheck.py halo_windows=1 amg_exact_sweeps=1 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5b.log 2>&1
cat gpurun_out/r5b.log | cut -c1-700
