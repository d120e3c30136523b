"""per rank-iteration summary of a rocprofv3 kernel_stats.csv of the team rehearsal (scripts/gpu_r4_team8_profile.sh)
usage: python scripts/prof_rank_iteration.py STATS.csv RANK_ITERATIONS"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = float(sys.argv[2])
setup = ("k_binv", "k_dinv", "k_lp_copies", "k_gj_", "k_galerkin", "k_fused_", "k_residual_tet", "k_gather_residual", "k_dense_pad", "k_dense_to",
         "k_bsr_to_dense", "k_fill_slot", "k_empty_coarse", "k_element", "k_multi_", "k_scale_by_rsqrt", "k_fill_pattern", "k_rep_scatter")
cat = {}
for r in rows:
    n, t, c = r["Name"], float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    key = ("setup / assembly / estimates" if any(s in n for s in setup)
           else "device copies / memsets" if ("copyBuffer" in n or "fillBuffer" in n) else "Krylov loop + V-cycle kernels")
    cat.setdefault(key, [0.0, 0])
    cat[key][0] += t
    cat[key][1] += c
for k, v in cat.items():
    print(f"{k:32s} {v[0] / N * 1e3:7.1f} us in {v[1] / N:5.1f} launches per rank-iteration")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    n = r["Name"].replace("void sns::", "").replace("sns::", "")[:60]
    print(f"  {n:62s} {int(r['Calls']) / N:8.2f} calls  {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / 1e3 / N:8.1f} us per rank-iteration")
