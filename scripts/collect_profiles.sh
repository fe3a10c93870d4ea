#!/bin/bash
# Runs ON the GPU box (gpurun -- bash scripts/collect_profiles.sh <outdir> [bench args...]): the bench line + breakdown, a
# single-stream rocprofv3 kernel trace, and the two PMC passes (one counter each, no other trace domain) of the same workload.
# rocprofv3 gets the program directly after `--` (python3 ...), as the pool requires.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
if [ -z "$ONLY_PMC" ]; then      # (ONLY_PMC=1: just the two counter passes, into a directory that already holds the rest)
python3 $GRAFT_REPO_ROOT/bench.py "$@" --breakdown $OUT/breakdown.txt > $OUT/bench.log 2> $OUT/bench.err || exit 1
echo "bench done" > $OUT/progress
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --inproc --serial --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/serial.log 2>&1 || exit 2
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
echo "trace done" >> $OUT/progress
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --inproc --serial --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_fetch.log 2>&1 || exit 3
find /tmp/prof_f -name "*counter_collection.csv" -exec cp {} $OUT/fetch_counter_collection.csv \;
echo "fetch done" >> $OUT/progress
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --inproc --serial --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_write.log 2>&1 || exit 4
find /tmp/prof_w -name "*counter_collection.csv" -exec cp {} $OUT/write_counter_collection.csv \;
echo "write done" >> $OUT/progress
# keep only the per-dispatch counter rows of the GEMM kernel (the merged-back directory is capped at 64 MiB)
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
for tag, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    per, kn = collections.defaultdict(float), {}
    with open(f"{out}/{tag}_counter_collection.csv") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == name:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
                kn[r["Dispatch_Id"]] = r["Kernel_Name"]
    with open(f"{out}/{tag}_per_dispatch.csv", "w") as fh:
        fh.write("dispatch,kernel,value_kb\n")
        for k, v in per.items():
            name = kn[k].replace("void ", "").replace("(anonymous namespace)::", "")
            short = "koaf_gemm_kernel" if "koaf_gemm_kernel" in name else name.split("(")[0].split("<")[0][-60:]     # (wgrad3x3_ring_kernel keeps its own name)
            fh.write(f"{k},{short},{v}\n")
import os
for tag in ("fetch", "write"):
    os.remove(f"{out}/{tag}_counter_collection.csv")
PY
