#!/bin/bash
# One-rank RCCL rehearsal of the HEADLINE workload on a 1-GPU box (gpurun -- bash scripts/rehearse_headline_rccl.sh <outdir>): the
# process group is built with backend nccl (= RCCL) in a world of one and every collective of the N > 1 path runs -- parameter
# broadcast, bucketed all-reduce of the 1.56 GB gradient arena launched from the backward hooks, barrier, gather of the per-rank
# times -- plus the host enqueue times of the steps a rank issues.  The log goes to profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
KOAF_DIST_REHEARSAL=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $OUT/rccl_rehearsal_syn3.log 2> $OUT/rccl_rehearsal_syn3.err || exit 1
timeout -k 10 200 python3 scripts/host_time.py native3 > $OUT/host_time_native3.log 2>&1 || exit 2
timeout -k 10 300 python3 scripts/host_time.py syn3 > $OUT/host_time_syn3.log 2>&1 || exit 3
