"""Keep only the PMC rows of the level-0 SpMV family and the assembly kernels (the full CSV is tens of MB).
usage: trim_pmc.py <in counter_collection.csv> <out csv>"""
import csv, re, sys
keep = ("k_fused_", "k_element<", "k_gather_", "k_residual_tet")
fine = re.compile(r"k_spmv(_lp)?<\d+, 1, |k_post_lp<\d+, 1>")     # FINE == 1 instantiations (level 0)
with open(sys.argv[1]) as fi, open(sys.argv[2], "w", newline="") as fo:
    r = csv.DictReader(fi)
    w = csv.DictWriter(fo, fieldnames=r.fieldnames)
    w.writeheader()
    for row in r:
        k = row.get("Kernel_Name", "")
        if any(s in k for s in keep) or fine.search(k):
            w.writerow(row)
