#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer.py -x -q -s -m gpu -k "window_cycle or peer or bench" > gpurun_out/r5j_tests.log 2>&1; echo "pytest rc $?"; grep -v amdgpu.ids gpurun_out/r5j_tests.log | tail -14 | cut -c1-400
timeout -k 10 900 python scripts/gpu_r5_pin_variants.py 4,8 > gpurun_out/r5j_pin.log 2>&1; grep -v amdgpu.ids gpurun_out/r5j_pin.log | cut -c1-300
