// Experiment harness (NOT part of the C-ABI of include/sns.h, not in the shipped libsns.so): built only with
// `make -C csrc HARNESS=1` (-DSNS_HARNESS).  It adds
//   * sns_bench_variants: interleaved A/B timing of two kernel variants in one process; ms_out[v] = average launch ms
//       which 0 = fp64 y=Ax default loads vs non-temporal matrix loads,
//       which 3 = fp64 y=Ax production (first 16 blocks of a row requested up-front) vs the stepped loop it replaced,
//       which 1 = low-precision Jacobi sweep fp16 row-scaled vs fp32 (both copies must exist: SNS_BOTH_LP=1 at pc_setup)
//   * the extra template instantiations those variants need (k_spmv<.,.,0|2|3,.>, k_spmv_lp<.,1,0,2,0>)
//   * the in-solver switches SNS_FP64_STEPPED / SNS_LP_STEPPED / SNS_BOTH_LP / SNS_GRAPH_ROWS (read per launch)
// used by scripts/gpu_ab*.py, gpu_r2_win_ab.py, gpu_r2_fp64_upfront_ab.py, gpu_r2_insolver_ab.sh, gpu_r2_coarse_sweep.sh.
#pragma once
#include "sns.h"
#ifdef SNS_HARNESS
extern "C" int sns_bench_variants(sns_handle h, int which, int rounds, int reps, double ms_out[2]);
#endif
