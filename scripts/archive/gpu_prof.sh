#!/bin/bash
# kernel-trace profile of one bench step (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-r01}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT.bench.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT.kernel_stats.csv
find $OUT -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'head -1 {} > '$OUT'.kernel_trace_head.csv; grep -m 3 "k_mm_live<4, 4, false" {} >> '$OUT'.kernel_trace_head.csv'
rm -rf $OUT
ls -la $GRAFT_REPO_ROOT/gpurun_out/
