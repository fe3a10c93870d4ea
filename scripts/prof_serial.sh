set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python3 $GRAFT_REPO_ROOT/bench.py --inproc --serial --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/serial.log 2>&1 || exit 2
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
