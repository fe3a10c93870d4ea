"""Copy one gpurun_out/<dir> collection (bench.log, breakdown.txt, serial/s_kernel_stats.csv, pmc_fetch, pmc_write) into
profiles/ and print the figures profiles/README.md quotes.  Usage: python scripts/install_profiles.py gpurun_out/final5"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

O = Path(sys.argv[1])
ROOT = Path(__file__).resolve().parent.parent
d = json.loads([ln for ln in open(O / "bench.log", errors="ignore") if ln.startswith('{"metric"')][-1])


def load(f, name):
    per, kn = collections.defaultdict(float), {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            kn[r["Dispatch_Id"]] = r["Kernel_Name"]
    return per, kn


fe, kn = load(O / "pmc_fetch" / "f_counter_collection.csv", "FETCH_SIZE")
wr, kn2 = load(O / "pmc_write" / "w_counter_collection.csv", "WRITE_SIZE")
gf = [v for k, v in fe.items() if "koaf_gemm_kernel" in kn[k]]
gw = [v for k, v in wr.items() if "koaf_gemm_kernel" in kn2[k]]
assert len(gf) == len(gw)
n = len(gf)
fetch_b, write_b = sum(gf) * 1024 * 2, sum(gw) * 1024
out = {
    "kernel": "koaf_gemm_kernel (all instantiations)",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one counter per pass) --kernel-trace --output-format csv -- python3 "
               "bench.py --serial --batch 8 --steps 2 --warmup 1 --no-cpu-baseline",
    "launches": n, "FETCH_SIZE_KB_sum": round(sum(gf), 1), "WRITE_SIZE_KB_sum": round(sum(gw), 1),
    "corrections": "bytes = FETCH_SIZE*1024*2 (gfx950 reports half of 16-B/lane coalesced reads) + WRITE_SIZE*1024",
    "bytes_per_launch": round((fetch_b + write_b) / n), "read_bytes_per_launch": round(fetch_b / n),
    "write_bytes_per_launch": round(write_b / n),
    "note": "memory-side (fabric) requests of the L2s: Infinity-Cache hits are counted, so this is an upper bound on HBM bytes",
}
json.dump(out, open(ROOT / "profiles" / "r01_gemm_traffic.json", "w"), indent=1)
shutil.copy(O / "serial" / "s_kernel_stats.csv", ROOT / "profiles" / "r01_kernel_stats_native3_b8_serial.csv")
shutil.copy(O / "bench.log", ROOT / "profiles" / "r01_bench_native3_b8.log")
shutil.copy(O / "breakdown.txt", ROOT / "profiles" / "r01_gemm_breakdown_native3_b8.txt")
rows = list(csv.DictReader(open(O / "serial" / "s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
g = [r for r in rows if "koaf_gemm_kernel" in r["Name"]]
gt, gc = sum(float(r["TotalDurationNs"]) for r in g), sum(int(r["Calls"]) for r in g)
r = d["roofline"]
print(f"step {d['ms_per_step']} ms {d['value']} knees/s | pinned {d['pinned_reference_model']['ms_per_step']} ms "
      f"{d['pinned_reference_model']['value']} knees/s | gemm {r['achieved']} TF/s frac {r['frac']} of {r['peak']} "
      f"({r['achieved_over_fp32_mfma_peak']} of fp32 peak) over {r['kernel_ms_per_step']} ms, {r['launches_per_step']} calls | "
      f"cpu {d['cpu_baseline']['value']}")
print(f"serial rocprof: all {tot / 1e6 / 15:.1f} ms/step, gemm {gt / 1e6 / 15:.1f} ms/step, {gc / 15:.0f} launches/step, "
      f"avg {gt / gc / 1e3:.1f} us | traffic {out['bytes_per_launch'] / 1e6:.0f} MB/launch")
