#!/bin/bash
# round 3: the cap of the block-Jacobi damping (amg_omega, default 0.8; the per-level value is min(cap, 4 / (3 |lambda|max))) across workloads
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 3 --warmup 0 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:40s} {d['ms_per_step']:9.1f} ms/step its {its} stokes {d['config']['stokes_its']}", flush=True)
PY
}
for cap in 0.8 0.7 0.65 0.6 0.55; do
  run "headline cap $cap" --opt amg_omega=$cap
  run "160x40x40 cap $cap" --cells 160,40,40 --opt amg_omega=$cap
  run "200x50x50 cap $cap" --cells 200,50,50 --opt amg_omega=$cap
  run "400x100x100 cap $cap" --cells 400,100,100 --opt amg_omega=$cap
  run "config 3 cap $cap" --config 3 --opt amg_omega=$cap
  run "config 4 cap $cap" --config 4 --opt amg_omega=$cap
done
