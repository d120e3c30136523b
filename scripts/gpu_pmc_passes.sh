#!/bin/bash
# PMC passes over scripts/gpu_pmc_spmv.py (one counter set per process, kernel-trace only).
# NOTE: TA_* / TCP_* counters abort rocprofv3 on this pool (signal 6, then a hang) -- SQ_* and TCC_* sets only.
R=$(pwd)
out=$R/gpurun_out/pmc_spmv
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 $R/scripts/gpu_pmc_spmv.py > $out/p$i.log 2>&1 || { echo "pass $i failed: $set"; break; }
  echo "pass $i done: $set"
done
cd $R
python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_spmv/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_spmv<0, 1" in k: name = "fp64_ax"
        elif "k_spmv_lp<2, 1" in k: name = "lp_jacobi_fine"
        elif "k_spmv_lp<1, 1" in k: name = "lp_resid_fine"
        elif "k_spmv_lp<2, 0" in k: name = "lp_jacobi_coarse"
        elif "k_fused_offdiag" in k: name = "fused_offdiag"
        else: continue
        e = d[(name, r["Counter_Name"])]
        e[0] += float(r["Counter_Value"]); e[1] += 1
with open("gpurun_out/pmc_spmv/summary.txt", "w") as fo:
    for (n, c), (s, k) in sorted(d.items()):
        fo.write(f"{n:18s} {c:40s} avg {s / k:16.1f}  (n={k})\n")
print(open("gpurun_out/pmc_spmv/summary.txt").read())
PY
rm -rf $out/p*/
