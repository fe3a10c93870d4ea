#!/bin/bash
# Runs ON the GPU box: two PMC passes (SQ counters do not fit one) over a GEMM launch script -- scripts/bench_gemm_pmc.py (the
# 3x3 256->256 layer through every operand path; default) or, with a second argument `short`, scripts/bench_gemm_pmc_short.py (the
# 64- and 128-channel 3x3 layers + the 1x1 convolutions around them, which also writes its launch plan).  The per-dispatch
# counter CSVs are reduced to per-kernel averages by scripts/sq_counters_summary.py (run in the build container, on the merged-back files).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
PROG=$GRAFT_REPO_ROOT/scripts/bench_gemm_pmc.py
ARGS="20"
if [ "$2" = "short" ]; then PROG=$GRAFT_REPO_ROOT/scripts/bench_gemm_pmc_short.py; ARGS="12 $OUT/plan.json"; fi
if [ "$2" = "t2d" ]; then PROG=$GRAFT_REPO_ROOT/scripts/bench_t2d_pmc.py; ARGS="12 $OUT/plan.json"; fi            # (round 4: rectangle-tile 3x3 kernel vs the raster halo kernel)
if [ "$2" = "stream" ]; then PROG=$GRAFT_REPO_ROOT/scripts/bench_stream_pmc.py; ARGS="10 $OUT/plan.json"; fi   # (round 4: streamed 1x1 kernels vs the block-wide loader)
if [ "$2" = "new" ]; then PROG=$GRAFT_REPO_ROOT/scripts/bench_new_kernels_pmc.py; ARGS="10 $OUT/plan.json"; fi      # (the round-3 ring / stem / attention kernels)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq1 -o a -- python3 $PROG $ARGS > $OUT/pass1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/sq2 -o b -- python3 $PROG $ARGS > $OUT/pass2.log 2>&1 || exit 2
for t in a b; do
  find /tmp/sq1 /tmp/sq2 -name "${t}_counter_collection.csv" -exec cp {} $OUT/${t}_counter_collection.csv \;
  find /tmp/sq1 /tmp/sq2 -name "${t}_kernel_trace.csv" -exec cp {} $OUT/${t}_kernel_trace.csv \;
done
ls -la $OUT
