"""per-kernel averages of a rocprofv3 counter_collection CSV: python scripts/pmc_avg.py file.csv [substring]"""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in k: continue
        rows[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    print(k)
    for c, v in sorted(d.items()): print(f"    {c:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")
