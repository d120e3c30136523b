#!/bin/bash
# round 4, call o: the aggregates' inverse blocks in fp16 (with the fp16 matrix copies): parity tests, then block smoothing on the LARGE levels again
timeout -k 10 600 python -m pytest tests/test_gpu_amg.py -x -q -s > gpurun_out/r4o_amg_tests.log 2>&1; tail -8 gpurun_out/r4o_amg_tests.log | cut -c1-250
run() {
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} levels {d['config']['amg_levels']}", flush=True)
PY
}
T="--steps 8 --warmup 2"
SLAB="--steps 8 --warmup 2 --cells 38,75,75 --length 0.5"
for rep in 1 2 3; do
run "10M default (block <= 8192 rows)" $T
run "10M block everywhere, l2=3" $T --opt amg_block_max_rows=0
run "10M block everywhere, l2=4" $T --opt amg_block_max_rows=0 --opt amg_bnu_l2=4
run "10M block <= 32768, l2=4" $T --opt amg_block_max_rows=32768 --opt amg_bnu_l2=4
done
for rep in 1 2; do
run "slab default" $SLAB
run "slab block <= 32768" $SLAB --opt amg_block_max_rows=32768
done
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 block everywhere" --config 4 --steps 4 --warmup 1 --opt amg_block_max_rows=0
run "cfg4u default" --config 4u --steps 4 --warmup 1
run "24M default" --cells 400,100,100 --steps 3 --warmup 1
run "24M block everywhere" --cells 400,100,100 --steps 3 --warmup 1 --opt amg_block_max_rows=0
