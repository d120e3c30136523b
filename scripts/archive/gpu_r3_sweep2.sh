#!/bin/bash
# round 3: spectral estimates on the smoother's own (fp16) matrix copy; level-1 1+6 sweeps as the single-GPU default
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 4 "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print(f"{sys.argv[1]:44s} {d['ms_per_step']:8.2f} ms  its {[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]} stokes {d['config']['stokes_its']} {d['config']['phase_ms_per_step']}", flush=True)
PY
}
for rep in 1 2; do
run "default"
run "L1 1+6" --opt amg_nu_l1_pre=1 --opt amg_nu_l1_post=6
run "L1 1+6 config 3" --config 3 --opt amg_nu_l1_pre=1 --opt amg_nu_l1_post=6
run "default config 3" --config 3
run "L1 1+6 config 4" --config 4 --opt amg_nu_l1_pre=1 --opt amg_nu_l1_post=6
run "default config 4" --config 4
done
