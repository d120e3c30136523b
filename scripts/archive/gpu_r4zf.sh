#!/bin/bash
# round 4, call zf: bench.py --gpus N end to end on the 1-GPU box -- N processes over peer windows sharing cuda:0 (--transport peer --shared-gpu):
# launch, slab partition, timed steps, weak leg, one JSON line.  Timings mean nothing (the ranks compete for one GPU).
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 4; do
timeout -k 10 800 python bench.py --gpus $N --transport peer --shared-gpu --steps 2 --warmup 1 --weak-timeout 400 > gpurun_out/r4zf_bench_shared_N$N.json 2> gpurun_out/r4zf_bench_shared_N$N.err; echo "N=$N rc $?"
python - $N <<'PY'
import json, sys
N = sys.argv[1]
lines = [l for l in open(f"gpurun_out/r4zf_bench_shared_N{N}.json") if l.startswith("{")]
if not lines:
    print("no JSON line"); print(open(f"gpurun_out/r4zf_bench_shared_N{N}.err").read()[-1500:]); sys.exit(0)
d = json.loads(lines[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "transport", "shared_gpu_rehearsal", "degraded", "scaling")})
print(d["config"]["parallelism"], [b for a, b, c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["krylov_loop_last_solve"], d["halo_overlap_selfcheck"])
print("weak:", d["weak_scaling"], "peer leg:", d["peer_transport"])
PY
done
