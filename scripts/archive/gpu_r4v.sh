#!/bin/bash
# round 4, call v: bench.py's peer-transport leg rehearsed with ONE rank (SNS_FORCE_DIST=1: the partitioned code path, RCCL communicator
# of one rank for the headline, then the same steps over a one-rank peer window), and the default N = 1 line beside it (must be untouched)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
SNS_FORCE_DIST=1 timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-f64-rerun --no-cpu-baseline --no-weak > gpurun_out/r4v_bench_force_dist.json 2> gpurun_out/r4v_bench_force_dist.err || { tail -20 gpurun_out/r4v_bench_force_dist.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4v_bench_force_dist.json"))
print({k: d[k] for k in ("value", "ms_per_step", "transport", "n_gpus", "degraded")}, d.get("peer_transport"), d.get("rccl_transport"))
PY
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-f64-rerun --no-cpu-baseline > gpurun_out/r4v_bench_n1.json 2> gpurun_out/r4v_bench_n1.err || { tail -20 gpurun_out/r4v_bench_n1.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4v_bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step", "transport", "n_gpus", "degraded")}, d.get("peer_transport"), d["config"]["phase_ms_per_step"])
PY
