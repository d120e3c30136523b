#!/bin/bash
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 3 --warmup 0 --cells $1 "${@:3}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 $2 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1 $2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:44s} {d['ms_per_step']:9.1f} ms/step its {its} stokes {d['config']['stokes_its']} levels {d['config']['amg_levels']}", flush=True)
PY
}
for c in 200,50,50 240,60,60 160,40,40; do
run $c "default"
run $c "L1 4+4" --opt amg_nu_l1_pre=4 --opt amg_nu_l1_post=4
run $c "unfused" --opt amg_fused_post=0
run $c "L1 4+4 unfused" --opt amg_nu_l1_pre=4 --opt amg_nu_l1_post=4 --opt amg_fused_post=0
done
