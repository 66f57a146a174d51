#!/bin/bash
# kernel-time breakdown of prof_small.py "$@" ; TAG env names the output
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
OUT=$R/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/prof_small.py "$@" > $OUT.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT.kernel_stats.csv
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
cp $f $OUT.kernel_trace_full.csv
rm -rf $OUT
grep "K=" $OUT.log | cut -c1-150
python3 - $OUT.kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:10.1f} ms  avg {float(r['AverageNs'])/1e3:10.1f} us  {float(r['Percentage']):6.2f}%")
PY
