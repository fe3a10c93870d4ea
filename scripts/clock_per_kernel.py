"""(GPU box, after a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv` pass) shader clock and
matrix-pipe occupancy per kernel, from the per-dispatch counter CSV: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, busy =
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clock); kernels grouped by name (GEMM: by template arguments), the first two
launches of a group skipped.
    python scripts/clock_per_kernel.py <counter_collection.csv> <out.json> [min total ms to list = 5]"""
import collections
import csv
import json
import re
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))
name, dur = {}, {}
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        m = re.search(r"(koaf_gemm_kernel<[^>]*>)", n)
        name[d] = m.group(1) if m else n.split("(")[0][:80]
        dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # us
groups = collections.defaultdict(list)
for d in sorted(per, key=int):
    groups[name[d]].append(d)
MIN_MS = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
out = []
for k, ds in groups.items():
    ds = ds[2:] if len(ds) > 4 else ds
    t = sum(dur[d] for d in ds)
    if t / 1e3 < MIN_MS:
        continue
    gui = sum(per[d]["GRBM_GUI_ACTIVE"] for d in ds)
    busy = sum(per[d]["SQ_VALU_MFMA_BUSY_CYCLES"] for d in ds)
    clock = gui / 8 / t / 1e3
    out.append({"kernel": k, "launches": len(ds), "total_ms": round(t / 1e3, 2), "avg_us": round(t / len(ds), 1), "clock_ghz": round(clock, 3),
                "mfma_busy_frac_at_clock": round(busy / 1024 / (t * clock * 1e3), 3) if clock else None})
out.sort(key=lambda e: -e["total_ms"])
json.dump({"note": "one rocprofv3 --pmc pass (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES); clock = GRBM_GUI_ACTIVE / 8 / duration", "kernels": out},
          open(sys.argv[2], "w"), indent=1)
for e in out[:25]:
    print(e)
