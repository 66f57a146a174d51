#!/bin/bash
# kernel-time breakdown of scripts/prof_small.py "$@" (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_small
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/prof_small.py "$@" > $OUT.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT.kernel_stats.csv
rm -rf $OUT
cut -d, -f1-5 $OUT.kernel_stats.csv | cut -c1-110 | head -14
