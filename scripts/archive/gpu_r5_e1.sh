#!/bin/bash
# round 5, experiment 1 (existing code, options only): iteration counts of candidate level-1 schedules of the 8-way strong split
# (team transport, 8 threads on one GPU).  amg_sweep_exchange_rows makes level 1's sweeps the exact global ones.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R() { echo "== $*"; timeout -k 10 300 python scripts/gpu_r4_strong_rehearsal.py 8 300,75,75 "$@" 2>&1 | grep "^N=" | cut -c1-330; }
{
R
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=3
R amg_nu_l1_pre=1 amg_nu_l1_post=3
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=2
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=1
R amg_nu_l1_pre=1 amg_nu_l1_post=1
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=2 amg_nu_l1_post=2
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=3 amg_bnu_l2=3
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=3 amg_bnu_l2=2
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=3 amg_dense_rows=768
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=3 amg_block_fine_rows=0
R amg_block_fine_rows=0
R amg_sweep_exchange_rows=30000 amg_nu_l1_pre=1 amg_nu_l1_post=3 amg_replicate_rows=0
} > gpurun_out/r5_e1.log 2>&1
cat gpurun_out/r5_e1.log
