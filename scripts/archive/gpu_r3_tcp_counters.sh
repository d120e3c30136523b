#!/bin/bash
# round 3: ONE TA / TCP counter per process over a small driver; stops at the first pass that fails (no retries)
R=$(pwd); out=$R/gpurun_out/pmc_tcp; mkdir -p $out
cells=${1:-140,35,35}
cd /tmp && export TMPDIR=/tmp
i=0
for c in TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 $R/scripts/gpu_r3_pmc_small.py $cells > $out/p$i.log 2>&1
  rc=$?
  echo "pass $i $c rc $rc"
  if [ $rc -ne 0 ]; then tail -15 $out/p$i.log; break; fi
done
cd $R
python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_tcp/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_spmv<0, 1" in k: name = "fp64_ax"
        elif "k_spmv_lp<1, 1" in k: name = "lp_resid_fine"
        elif "k_post_lp<2, 1" in k: name = "post_fine"
        elif "k_spmv_lp<2, 0" in k: name = "lp_jacobi_coarse"
        else: continue
        e = d[(name, r["Counter_Name"])]
        e[0] += float(r["Counter_Value"]); e[1] += 1
with open("gpurun_out/pmc_tcp/summary.txt", "w") as fo:
    for (n, c), (s, k) in sorted(d.items()):
        fo.write(f"{n:18s} {c:44s} avg {s / k:18.1f}  (n={k})\n")
print(open("gpurun_out/pmc_tcp/summary.txt").read())
PY
rm -rf $out/p*/
