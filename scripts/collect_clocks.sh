#!/bin/bash
# Runs ON the GPU box: the shader clock the kernels actually run at -- one PMC pass (GRBM_GUI_ACTIVE + matrix-pipe busy cycles) over
# (a) one single-stream headline step and (b) the sustained 256-row halo microbench (scripts/bench_halo256.py 1280 60), reduced on the
# box by scripts/clock_per_kernel.py.   gpurun -- bash scripts/collect_clocks.sh <outdir>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/clk_s -o s -- python3 $GRAFT_REPO_ROOT/bench.py --inproc --serial --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/step.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/scripts/clock_per_kernel.py $(find /tmp/clk_s -name "s_counter_collection.csv") $OUT/clocks_step.json 20 > $OUT/clocks_step.txt || exit 2
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/clk_m -o m -- python3 $GRAFT_REPO_ROOT/scripts/bench_halo256.py 1280 60 > $OUT/micro.log 2>&1 || exit 3
python3 $GRAFT_REPO_ROOT/scripts/clock_per_kernel.py $(find /tmp/clk_m -name "m_counter_collection.csv") $OUT/clocks_micro.json 20 > $OUT/clocks_micro.txt || exit 4
cat $OUT/clocks_step.txt $OUT/clocks_micro.txt
