#!/bin/bash
# kernel-time breakdown of any script: bash scripts/gpu_prof_any.sh scripts/prof_kmeans.py 397 1000 20
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_any
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/"$@" > $OUT.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT.kernel_stats.csv
rm -rf $OUT
tail -2 $OUT.log
cut -d, -f1-5 $OUT.kernel_stats.csv | sed 's/(.*"/"/' | head -8
