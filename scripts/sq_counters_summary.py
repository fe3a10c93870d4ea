"""Reduce the two rocprofv3 --pmc passes of scripts/collect_sq_counters.sh to per-kernel averages ->
profiles/<round>_gemm_sq_counters[_short].json.  Usage: python scripts/sq_counters_summary.py gpurun_out/<dir> [round tag]
(with a plan.json in the directory -- the `short` collection -- each kernel template's dispatches are split, in launch order,
among the plan's entries that use it)
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import json
import re
import sys
from pathlib import Path

O = Path(sys.argv[1])
RND = sys.argv[2] if len(sys.argv) > 2 else "r03"
ROOT = Path(__file__).resolve().parent.parent
PLAN = json.loads((O / "plan.json").read_text()) if (O / "plan.json").exists() else None
BYNAME = PLAN is not None and "kernel" in PLAN[0]       # the `new` collection: plan entries name a kernel, not a GEMM template
LABEL = {
    "128, 128, 0, 0, 0, 0, true, false, 256": "dense forward 4096^3, bf16 x 3 (6 MFMAs per product)",
    "128, 128, 1, 6, 1, 0, true, true, 256": "conv 3x3 256->256 forward, fp16 x 2 (3 MFMAs), fp32 activation loader: BN prologue + split in the k-loop, weight tiles by LDS-DMA",
    "128, 128, 7, 6, 0, 0, true, true, 256": "conv 3x3 256->256 forward, activation plane images gathered per tap by LDS-DMA (both operands DMA)",
    "256, 128, 9, 6, 0, 0, true, true, 512": "conv 3x3 256->256 forward, HALO kernel: 256-pixel raster tile + halo resident in LDS for all nine taps, 8 waves",
    "128, 128, 2, 6, 2, 0, true, true, 256": "conv 3x3 256->256 data gradient, fp32 loader: dy formed from (dz, c) on load, weight tiles by LDS-DMA",
    "128, 128, 3, 4, 2, 1, true, true, 256": "conv 3x3 256->256 weight gradient, fp32 loaders: dy formed on load, x through its BN prologue",
    "128, 128, 10, 11, 0, 0, true, true, 256": "conv 3x3 256->256 weight gradient, both operands from plane images (K-major LDS-DMA)",
    "128, 128, 1, 0, 1, 0, true, false, 256": "conv 3x3 256->256 forward, bf16 x 3 (the round-1 scheme)",
}


def load(tag):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name, dur = {}, {}
    with open(O / f"{tag}_counter_collection.csv") as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"koaf_gemm_kernel<([^>]*)>", r["Kernel_Name"])
            if not m and BYNAME:
                m = re.search(r"(wgrad3x3_ring_kernel|stem_fwd_mma_kernel|stem_wgrad_mma_kernel|attention_fwd_kernel)", r["Kernel_Name"])
            if not m:
                continue
            d = r["Dispatch_Id"]
            per[d][r["Counter_Name"]] += float(r["Counter_Value"])
            name[d] = m.group(1)
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return per, name, dur


out = []
pa, na, da = load("a")
pb, nb, db = load("b")


def series(key, per, names, idx=None, count=None):
    """counter dicts of the dispatches of template `key` (a prefix of the template argument list), in launch order; with a
    plan: the idx-th run of `count` launches of that template"""
    ds = sorted((d for d in per if names[d].startswith(key)), key=int)
    if idx is not None:
        ds = ds[idx * count:(idx + 1) * count]
    return ds[2:]                                    # (the first launches of a series warm the caches / clocks)


if PLAN is not None:
    items, seen = [], collections.Counter()
    for e in PLAN:
        key = e["kernel"] if BYNAME else e["template"]
        items.append((key, e["label"], seen[key], e["launches"]))
        seen[key] += 1
else:
    items = [(k, lab, None, None) for k, lab in LABEL.items()]
for key, label, idx, count in items:
    A = [pa[d] for d in series(key, pa, na, idx, count)]
    B = [pb[d] for d in series(key, pb, nb, idx, count)]
    us = [da[d] for d in series(key, pa, na, idx, count)]
    if not A or not B:
        print("no dispatches for", key, label)
        continue
    avg = lambda L, c: sum(x[c] for x in L) / len(L)      # noqa: E731
    t_us = sum(us) / len(us)
    clock = avg(A, "GRBM_GUI_ACTIVE") / 8 / t_us / 1e3    # GHz
    mfma = avg(B, "SQ_INSTS_MFMA") if avg(B, "SQ_INSTS_MFMA") else (avg(B, "SQ_INSTS_VALU_MFMA_MOPS_F16") + avg(B, "SQ_INSTS_VALU_MFMA_MOPS_BF16")) / 512
    wave = avg(A, "SQ_WAVE_CYCLES")
    out.append({
        "kernel": label, "template": key if BYNAME else f"koaf_gemm_kernel<{key}>", "us": round(t_us, 1), "clock_ghz": round(clock, 2),
        "mfma_busy_frac_at_clock": round(avg(A, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (t_us * clock * 1e3), 3),
        "valu_per_mfma": round((avg(A, "SQ_INSTS_VALU") - mfma) / mfma, 2),
        "lds_per_mfma": round(avg(B, "SQ_INSTS_LDS") / mfma, 2), "vmem_rd_per_mfma": round(avg(B, "SQ_INSTS_VMEM_RD") / mfma, 3),
        "salu_per_mfma": round(avg(B, "SQ_INSTS_SALU") / mfma, 2),
        "wait_any": round(avg(A, "SQ_WAIT_ANY") / wave, 3), "wait_inst_any": round(avg(A, "SQ_WAIT_INST_ANY") / wave, 3),
        "active_inst_any": round(avg(A, "SQ_ACTIVE_INST_ANY") / wave, 3),
    })
doc = {"command": "bash scripts/collect_sq_counters.sh <dir> [short] (two rocprofv3 --pmc passes, --kernel-trace --output-format csv, no other "
                  "trace domain, over scripts/bench_gemm_pmc.py 20 | scripts/bench_gemm_pmc_short.py 12 | scripts/bench_new_kernels_pmc.py 10), reduced by scripts/sq_counters_summary.py",
       "note": "averages over the launches of each kernel but its first two, on random operands; *_per_mfma are wave-instruction counts per matrix instruction; "
               "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clock); wait_* / active_* are shares of SQ_WAVE_CYCLES",
       "kernels": out}
fname = sys.argv[3] if len(sys.argv) > 3 else f"{RND}_sq_counters_new_kernels.json" if BYNAME else (f"{RND}_gemm_sq_counters_short.json" if PLAN is not None else f"{RND}_gemm_sq_counters.json")
json.dump(doc, open(ROOT / "profiles" / fname, "w"), indent=1)
for k in out:
    print(k)
