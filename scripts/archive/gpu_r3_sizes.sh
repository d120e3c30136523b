#!/bin/bash
# round 3: iterations and step time against the mesh size (one GPU, Re 200, first two Newton steps from the Stokes solution)
run() {
  python bench.py --no-cpu-baseline --no-f64-rerun --steps 2 --warmup 0 --cells $1 > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:14s} {d['config']['workload'].split('=')[1].split(',')[0].strip():>14s} {d['ms_per_step']:9.1f} ms/step {d['value']:6.1f} M-DOF/s  its {its} stokes {d['config']['stokes_its']} levels {d['config']['amg_levels']} ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.2f}", flush=True)
PY
}
for c in 100,25,25 200,50,50 300,75,75 400,100,100 480,120,120 600,150,150; do run $c; done
