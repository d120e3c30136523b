#!/bin/bash
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 2 --warmup 0 --cells $1 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1 ${*:2}" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:48s} {d['ms_per_step']:9.1f} ms/step {d['value']:6.1f} M-DOF/s  its {its} stokes {d['config']['stokes_its']} levels {d['config']['amg_levels']}", flush=True)
PY
}
for c in 300,75,75 340,85,85 400,100,100 480,120,120 600,150,150; do run $c; run $c --opt amg_nu_scale_with_size=0; done
