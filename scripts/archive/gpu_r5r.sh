#!/bin/bash
# round 5, call r5r: the replicated tail started one level deeper (the global 30 k-row level stays partitioned, exact sweeps)
mkdir -p gpurun_out
bash scripts/gpu_r5_team8_profile.sh r5r 8 amg_replicate_rows=8000 > gpurun_out/r5r_team8.log 2>&1; grep -A22 "rank-iterations" gpurun_out/r5r_team8.log | cut -c1-200
