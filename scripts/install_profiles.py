"""Copy one gpurun_out/<dir> collection made by scripts/collect_profiles.sh (bench.log, breakdown.txt, kernel_stats.csv,
fetch/write_per_dispatch.csv) into profiles/ as r02_* files and print the figures profiles/README.md quotes.
Usage: python scripts/install_profiles.py gpurun_out/<dir> <workload tag, e.g. syn3_b8>"""
import csv
import json
import shutil
import sys
from pathlib import Path

O, tag = Path(sys.argv[1]), sys.argv[2]
ROOT = Path(__file__).resolve().parent.parent
d = json.loads([ln for ln in open(O / "bench.log", errors="ignore") if ln.startswith('{"metric"')][-1])
workload = tag.split("_")[0]


def gemm_sum(f):
    vals = [float(r["value_kb"]) for r in csv.DictReader(open(f)) if r["kernel"] == "koaf_gemm_kernel"]
    return sum(vals), len(vals)


fe, n = gemm_sum(O / "fetch_per_dispatch.csv")
wr, n2 = gemm_sum(O / "write_per_dispatch.csv")
assert n == n2 and n > 0
fetch_b, write_b = fe * 1024 * 2, wr * 1024
B = int(d["config"]["workload"].split("per-GPU batch ")[1].split(",")[0])
out = {
    "kernel": "koaf_gemm_kernel (all instantiations)", "workload": d["config"]["workload"], "batch": B,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one counter per pass) --kernel-trace --output-format csv -- python3 "
               "bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline --no-secondary (scripts/collect_profiles.sh)",
    "launches": n, "FETCH_SIZE_KB_sum": round(fe, 1), "WRITE_SIZE_KB_sum": round(wr, 1),
    "corrections": "bytes = FETCH_SIZE*1024*2 (gfx950 reports half of 16-B/lane coalesced reads) + WRITE_SIZE*1024",
    "bytes_per_launch": round((fetch_b + write_b) / n), "read_bytes_per_launch": round(fetch_b / n),
    "write_bytes_per_launch": round(write_b / n),
    "note": "memory-side (fabric) requests of the L2s: Infinity-Cache hits are counted, so this is an upper bound on HBM bytes",
}
json.dump(out, open(ROOT / "profiles" / f"r02_gemm_traffic_{workload}.json", "w"), indent=1)
shutil.copy(O / "kernel_stats.csv", ROOT / "profiles" / f"r02_kernel_stats_{tag}_serial.csv")
shutil.copy(O / "bench.log", ROOT / "profiles" / f"r02_bench_{tag}.log")
shutil.copy(O / "breakdown.txt", ROOT / "profiles" / f"r02_gemm_breakdown_{tag}.txt")
rows = list(csv.DictReader(open(O / "kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
g = [r for r in rows if "koaf_gemm_kernel" in r["Name"]]
gt, gc = sum(float(r["TotalDurationNs"]) for r in g), sum(int(r["Calls"]) for r in g)
nsteps = 1 + 3 + 1 + 1      # warm-up + timed + allocator-settling + instrumented (bench.py under collect_profiles.sh)
r = d["roofline"]
print(f"step {d['ms_per_step']} ms (median {d['ms_per_step_median']}) {d['value']} knees/s | gemm {r['achieved']} TF/s frac {r['frac']} "
      f"of {r['peak']} over {r['kernel_ms_per_step']} ms, {r['launches_per_step']} calls | cpu {d.get('cpu_baseline', {}).get('value')}")
print(f"serial rocprof: all {tot / 1e6 / nsteps:.1f} ms/step, gemm {gt / 1e6 / nsteps:.1f} ms/step, {gc / nsteps:.0f} launches/step, "
      f"avg {gt / gc / 1e3:.1f} us | traffic {out['bytes_per_launch'] / 1e6:.0f} MB/launch "
      f"(algorithmic {r['algorithmic_gbytes_per_step'] * 1e3 / r['launches_per_step']:.0f} MB/launch)")
