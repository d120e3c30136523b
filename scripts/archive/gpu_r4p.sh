#!/bin/bash
# round 4, call p: candidate defaults A (aggregate blocks on every coarse level, 4 + 4 on level 2) against B (3 + 3) and against the size-limited policy
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
B="--opt amg_bnu_l2=3"
S="--opt amg_block_max_rows=8192 --opt amg_bnu_l2=3"
for rep in 1 2; do
run "10M A (new default)" $T
run "10M B (l2 = 3)" $T $B
run "10M size-limited blocks (r4j)" $T $S
run "slab A" $SLAB
run "slab B" $SLAB $B
run "slab size-limited" $SLAB $S
done
run "cfg3 A" --config 3 --steps 8 --warmup 2
run "cfg3 B" --config 3 --steps 8 --warmup 2 $B
run "cfg4 A" --config 4 --steps 4 --warmup 1
run "cfg4 B" --config 4 --steps 4 --warmup 1 $B
run "cfg4 size-limited" --config 4 --steps 4 --warmup 1 $S
run "cfg4u A" --config 4u --steps 4 --warmup 1
run "cfg4u B" --config 4u --steps 4 --warmup 1 $B
run "cfg4u size-limited" --config 4u --steps 4 --warmup 1 $S
run "24M A" --cells 400,100,100 --steps 3 --warmup 1
run "81M A" --cells 600,150,150 --steps 3 --warmup 1
run "81M size-limited" --cells 600,150,150 --steps 3 --warmup 1 $S
timeout -k 10 600 python scripts/gpu_r4_hardcase.py > gpurun_out/r4p_hardcase.log 2>&1; grep "retry 0" gpurun_out/r4p_hardcase.log | cut -c1-330
timeout -k 10 300 python scripts/gpu_r4_unstructured.py 28 > gpurun_out/r4p_unstructured.log 2>&1; grep tets gpurun_out/r4p_unstructured.log | cut -c1-200
timeout -k 10 300 python scripts/gpu_r4_strong_rehearsal.py 1,8 > gpurun_out/r4p_strong.log 2>&1; grep "^N=" gpurun_out/r4p_strong.log | cut -c1-420
timeout -k 10 300 python scripts/gpu_weak_rehearsal.py 8 2>&1 | grep -E "^N=|owned nodes" | head -2 | cut -c1-220
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_2d.py::test_dfg2d_constants_on_the_3d_tet_path > gpurun_out/r4p_gputests.log 2>&1; tail -6 gpurun_out/r4p_gputests.log | cut -c1-200
