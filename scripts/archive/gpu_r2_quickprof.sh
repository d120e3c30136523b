#!/bin/bash
# kernel-trace stats of a short bench run: prints the top kernel rows (usage: bash scripts/gpu_r2_quickprof.sh tag [bench args])
tag=$1; shift
R=$(pwd); out=$R/gpurun_out/qp_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f64-rerun "$@" > $out/bench.json 2> $out/err.log
cd $R
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows)/1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:16]:
    print(f'{r["Name"][:60]:60s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us {float(r["TotalDurationNs"])/1e6:8.1f} ms  min {float(r["MinNs"])/1e3:7.1f} max {float(r["MaxNs"])/1e3:7.1f}')
PY
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().split(chr(10))[-1]); print(d['ms_per_step'], [b for a,b,c in d['config']['newton_log_fnorm_kspits_reason']], d['config']['phase_ms_per_step'])"
rm -rf $out/stats
