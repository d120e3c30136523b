#!/bin/bash
# round 4, call zb: the replicated tail from level 1 on (global 218 k rows: the single-GPU hierarchy below the fine level on every rank) at the
# 8-way strong split -- iterations, then the per-rank kernel statistics (team, one stream)
mkdir -p gpurun_out
timeout -k 10 400 python scripts/gpu_r4_strong_rehearsal.py 8 300,75,75 amg_replicate_rows=400000 > gpurun_out/r4zb_team.log 2>&1; grep "^N=" gpurun_out/r4zb_team.log | cut -c1-500 || tail -5 gpurun_out/r4zb_team.log
