#!/bin/bash
# round 3: full GPU test-suite, then the per-round profile (scripts/gpu_profile_round.sh), then ONE converged run of the
# CPU baseline (the reference's algorithm restated: tfqmr + block-Jacobi/ILU(0) on the host cores) on the 10.1 M-tet mesh
tag=$1
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/${tag}_gputests.log | tail -4 || exit 1
bash scripts/gpu_profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1 || { tail -20 gpurun_out/${tag}_profile.log; exit 1; }
tail -3 gpurun_out/${tag}_profile.log
python scripts/prof_top.py gpurun_out/prof_$tag/${tag}_bench_kernel_stats.csv 40
if [ "$2" = "cpu" ]; then
  python bench.py --steps 2 --warmup 1 --no-f64-rerun --cpu-maxit 12000 > gpurun_out/${tag}_cpu_converged.json 2> gpurun_out/${tag}_cpu_converged.err
  grep cpu_baseline gpurun_out/${tag}_cpu_converged.err | tail -5
fi
