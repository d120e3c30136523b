#!/bin/bash
# round 2: level-1 sweeps 1 + 6 (default since) vs 4 + 4 in the partitioned solver (N threads on one GPU, team transport)
for v in "0 0" "1 6"; do
  set -- $v
  echo "== amg_nu_l1_pre $1 amg_nu_l1_post $2"
  SNS_TEAM_OPTS="amg_nu_l1_pre=$1,amg_nu_l1_post=$2" timeout -k 10 500 python scripts/gpu_team_check.py "(200,50,50)" 2>&1 | grep -v amdgpu || exit 1
done
