#!/bin/bash
# PMC passes over scripts/gpu_pmc_spmv.py (one counter set per process, kernel-trace only)
R=$(pwd)
out=$R/gpurun_out/pmc_spmv
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "MemUnitStalled OccupancyPercent GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 $R/scripts/gpu_pmc_spmv.py > $out/p$i.log 2>&1 || echo "pass $i failed"
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
        elif "k_spmv_f32<2, 1" in k: name = "f32_jacobi_fine"
        elif "k_spmv_f32<1, 1" in k: name = "f32_resid_fine"
        else: continue
        e = d[(name, r["Counter_Name"])]
        e[0] += float(r["Counter_Value"]); e[1] += 1
with open("gpurun_out/pmc_spmv/summary.txt", "w") as fo:
    for (n, c), (s, k) in sorted(d.items()):
        fo.write(f"{n:18s} {c:40s} avg {s / k:16.1f}  (n={k})\n")
print(open("gpurun_out/pmc_spmv/summary.txt").read())
PY
rm -rf $out/p*/
