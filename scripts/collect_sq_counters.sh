#!/bin/bash
# Runs ON the GPU box: two PMC passes (SQ counters do not fit one) over scripts/bench_gemm_pmc.py; the per-dispatch counter CSVs
# are reduced to per-kernel averages by scripts/sq_counters_summary.py (run here, on the merged-back files).
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq1 -o a -- python3 $GRAFT_REPO_ROOT/scripts/bench_gemm_pmc.py 20 > $OUT/pass1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/sq2 -o b -- python3 $GRAFT_REPO_ROOT/scripts/bench_gemm_pmc.py 20 > $OUT/pass2.log 2>&1 || exit 2
for t in a b; do
  find /tmp/sq1 /tmp/sq2 -name "${t}_counter_collection.csv" -exec cp {} $OUT/${t}_counter_collection.csv \;
  find /tmp/sq1 /tmp/sq2 -name "${t}_kernel_trace.csv" -exec cp {} $OUT/${t}_kernel_trace.csv \;
done
ls -la $OUT
