#!/bin/bash
# round-2 starting point at the north-star shape: kernel-time breakdown of the full K=1000 bench shape
# (10 batches x 125 tasks, 20 x 1000) and PMC passes on a short K=1000 run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02_base}
OUT=$R/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/prof_small.py 1000 10 125 20 > $OUT.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT.kernel_stats.csv
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
head -1 $f > $OUT.kernel_trace_head.csv; grep -m 4 "k_mm_live<32" $f >> $OUT.kernel_trace_head.csv
rm -rf $OUT
cat $OUT.log | tail -3
cut -d, -f1-5 $OUT.kernel_stats.csv | cut -c1-120 | head -16
PMC_TAG=$TAG bash $R/scripts/gpu_pmc.sh 1000 2 125 3
