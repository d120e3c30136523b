#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4
timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 '{"halo_windows": 0, "amg_exact_sweeps": 0}'
timeout -k 10 250 python scripts/gpu_r5_peer_case.py duct 3
timeout -k 10 250 python scripts/gpu_r5_peer_case.py cavity 4 '{"halo_windows": 0, "amg_exact_sweeps": 0, "halo_overlap": 0}'
} > gpurun_out/r5c.log 2>&1
grep -v amdgpu.ids gpurun_out/r5c.log | cut -c1-900
