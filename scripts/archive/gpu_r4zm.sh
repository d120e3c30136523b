#!/bin/bash
# round 4, call zm: k_resid_restrict on the single-GPU FINE level too (amg_fuse_restrict = 2) against the tuned k_spmv_lp + k_restrict_blk
mkdir -p gpurun_out
run() {
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-f64-rerun "${@:2}" > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_tmp.err || { echo "$1 FAILED"; tail -5 gpurun_out/sweep_tmp.err; return; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
its=[b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']]
k=d['roofline']['fine_level_spmv_kernels'].get('b_minus_ax',{})
print(f"{sys.argv[1]:36s} {d['ms_per_step']:8.2f} ms  its {its} krylov ms/it {d['config']['phase_ms_per_step']['krylov']*len(its)/sum(its):.3f} {d['config']['phase_ms_per_step']} fine residual launch {k.get('avg_launch_ms')} ms", flush=True)
PY
}
{
T="--steps 8 --warmup 2"
for rep in 1 2; do
run "10M default" $T
run "10M fused fine residual too" $T --opt amg_fuse_restrict=2
done
run "cfg3 default" --config 3 --steps 8 --warmup 2
run "cfg3 fused fine residual too" --config 3 --steps 8 --warmup 2 --opt amg_fuse_restrict=2
run "cfg4 default" --config 4 --steps 4 --warmup 1
run "cfg4 fused fine residual too" --config 4 --steps 4 --warmup 1 --opt amg_fuse_restrict=2
} > gpurun_out/r4zm.log 2>&1
cat gpurun_out/r4zm.log
