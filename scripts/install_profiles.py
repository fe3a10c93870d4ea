"""Copy one gpurun_out/<dir> collection made by scripts/collect_profiles.sh (bench.log, breakdown.txt, kernel_stats.csv,
fetch/write_per_dispatch.csv) into profiles/ as <round>_* files and print the figures profiles/README.md quotes.
Usage: python scripts/install_profiles.py gpurun_out/<dir> <workload tag, e.g. syn3_b8> [round tag, default r03]"""
import csv
import json
import shutil
import sys
from pathlib import Path

O, tag = Path(sys.argv[1]), sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "r03"
ROOT = Path(__file__).resolve().parent.parent
d = json.loads([ln for ln in open(O / "bench.log", errors="ignore") if ln.startswith('{"metric"')][-1])
workload = tag.split("_")[0]


GEMM_KERNELS = ("koaf_gemm_kernel", "wgrad3x3_ring_kernel")      # what bench.py's GEMM-family brackets launch (koaf_gemm.hip, koaf_wgrad3.hip)


def kernel_sum(f, name):
    names = name if isinstance(name, tuple) else (name,)
    vals = [float(r["value_kb"]) for r in csv.DictReader(open(f)) if any(n in r["kernel"] for n in names)]
    return sum(vals), len(vals)


PMC_STEPS = 4       # steps bench.py executes under the PMC passes: 1 warm-up + 1 timed + 1 allocator-settling + 1 instrumented
fe, n = kernel_sum(O / "fetch_per_dispatch.csv", GEMM_KERNELS)
wr, n2 = kernel_sum(O / "write_per_dispatch.csv", GEMM_KERNELS)
assert n == n2 and n > 0
fetch_b, write_b = fe * 1024 * 2, wr * 1024
pfe, pn = kernel_sum(O / "fetch_per_dispatch.csv", "act_planes_kernel")
pwr, _ = kernel_sum(O / "write_per_dispatch.csv", "act_planes_kernel")
planes_b = pfe * 1024 * 2 + pwr * 1024
B = int(d["config"]["workload"].split("per-GPU batch ")[1].split(",")[0])
out = {
    "kernel": "koaf_gemm_kernel (all instantiations) + wgrad3x3_ring_kernel", "workload": d["config"]["workload"], "batch": B,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one counter per pass) --kernel-trace --output-format csv -- python3 "
               "bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline --no-secondary (scripts/collect_profiles.sh)",
    "launches": n, "FETCH_SIZE_KB_sum": round(fe, 1), "WRITE_SIZE_KB_sum": round(wr, 1),
    "corrections": "bytes = FETCH_SIZE*1024*2 (gfx950 reports half of 16-B/lane coalesced reads) + WRITE_SIZE*1024",
    "bytes_per_launch": round((fetch_b + write_b) / n), "read_bytes_per_launch": round(fetch_b / n),
    "write_bytes_per_launch": round(write_b / n),
    # per STEP, which is what compares with bench.py's algorithmic_gbytes_per_step (a per-launch average against a per-call
    # average hid 12 % in round 2: one convolution call is several launches), and the plane-image pre-passes the GEMM-family
    # brackets include
    "steps_in_the_pass": PMC_STEPS, "launches_per_step": round(n / PMC_STEPS, 1),
    "bytes_per_step": round((fetch_b + write_b) / PMC_STEPS),
    "act_planes_launches_per_step": round(pn / PMC_STEPS, 1), "act_planes_bytes_per_step": round(planes_b / PMC_STEPS),
    "algorithmic_bytes_per_step": round(d["roofline"]["algorithmic_gbytes_per_step"] * 1e9),
    "ratio_to_algorithmic": round((fetch_b + write_b) / PMC_STEPS / (d["roofline"]["algorithmic_gbytes_per_step"] * 1e9), 3),
    "ratio_to_algorithmic_with_act_planes": round((fetch_b + write_b + planes_b) / PMC_STEPS / (d["roofline"]["algorithmic_gbytes_per_step"] * 1e9), 3),
    "note": "memory-side (fabric) requests of the L2s: Infinity-Cache hits are counted, so this is an upper bound on HBM bytes",
}
json.dump(out, open(ROOT / "profiles" / f"{RND}_gemm_traffic_{workload}.json", "w"), indent=1)
shutil.copy(O / "kernel_stats.csv", ROOT / "profiles" / f"{RND}_kernel_stats_{tag}_serial.csv")
shutil.copy(O / "bench.log", ROOT / "profiles" / f"{RND}_bench_{tag}.log")
shutil.copy(O / "breakdown.txt", ROOT / "profiles" / f"{RND}_gemm_breakdown_{tag}.txt")
rows = list(csv.DictReader(open(O / "kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
g = [r for r in rows if any(n in r["Name"] for n in GEMM_KERNELS)]
gt, gc = sum(float(r["TotalDurationNs"]) for r in g), sum(int(r["Calls"]) for r in g)
nsteps = 1 + 3 + 1 + 1      # warm-up + timed + allocator-settling + instrumented (bench.py under collect_profiles.sh)
r = d["roofline"]
print(f"step {d['ms_per_step']} ms (median {d['ms_per_step_median']}) {d['value']} knees/s | gemm {r['achieved']} TF/s frac {r['frac']} "
      f"of {r['peak']} over {r['kernel_ms_per_step']} ms, {r['launches_per_step']} calls | cpu {d.get('cpu_baseline', {}).get('value')}")
print(f"serial rocprof: all {tot / 1e6 / nsteps:.1f} ms/step, gemm {gt / 1e6 / nsteps:.1f} ms/step, {gc / nsteps:.0f} launches/step, "
      f"avg {gt / gc / 1e3:.1f} us | traffic {out['bytes_per_step'] / 1e9:.0f} GB/step (+ {out['act_planes_bytes_per_step'] / 1e9:.0f} GB plane-image "
      f"pre-passes) vs algorithmic {r['algorithmic_gbytes_per_step']:.0f} GB/step = {out['ratio_to_algorithmic']} / {out['ratio_to_algorithmic_with_act_planes']}")
for row in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"   {float(row['TotalDurationNs']) / 1e6 / nsteps:8.1f} ms/step {int(row['Calls']) / nsteps:7.1f} launches  {row['Name'][:110]}")
