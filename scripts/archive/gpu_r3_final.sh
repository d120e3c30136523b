#!/bin/bash
# round 3, final state: per-round profile (r3b) + the secondary bench lines + the single-rank RCCL path of bench.py
tag=$1
bash scripts/gpu_profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1 || { tail -20 gpurun_out/${tag}_profile.log; exit 1; }
python scripts/prof_top.py gpurun_out/prof_$tag/${tag}_bench_kernel_stats.csv 24
for c in 3 4 4u; do
  python bench.py --config $c --no-cpu-baseline > gpurun_out/${tag}_bench_config$c.json 2> gpurun_out/${tag}_bench_config$c.err || echo "config $c FAILED"
done
SNS_FORCE_DIST=1 python bench.py --steps 2 --no-cpu-baseline --no-f64-rerun --no-weak > gpurun_out/${tag}_force_dist.json 2> gpurun_out/${tag}_force_dist.err || echo "force-dist FAILED"
python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
for f in (f"prof_{tag}/{tag}_bench_unprofiled.json", f"prof_{tag}/{tag}_bench_under_rocprof.json", f"{tag}_bench_config3.json", f"{tag}_bench_config4.json", f"{tag}_bench_config4u.json", f"{tag}_force_dist.json"):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().split("\n")[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], [b for a, b, c in d["config"]["newton_log_fnorm_kspits_reason"]], d["config"]["phase_ms_per_step"],
          "f64", (d.get("all_f64_preconditioner") or {}).get("ms_per_step"), d.get("rccl_ranks"), d.get("halo_overlap_selfcheck"))
PY
