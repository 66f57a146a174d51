#!/bin/bash
# rocprofv3 kernel-trace + stats of one bench step: scripts/gpu_prof_bench.sh [k1000|k100|k397_hard|fs_k1000]
# (k1000: the headline; the others: that workload alone).  Writes gpurun_out/prof_bench_<w>.{kernel_stats.csv,kernel_trace_head.csv,bench.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-k1000}
OUT=$R/gpurun_out/prof_bench_$W
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | grep "^{" | tail -1 > $OUT.bench.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT.kernel_stats.csv
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
head -1 $f > $OUT.kernel_trace_head.csv; grep -m 3 "k_mm_live<.*false" $f >> $OUT.kernel_trace_head.csv; grep -m 3 "k_mm_split<" $f >> $OUT.kernel_trace_head.csv
rm -rf $OUT
python3 - $OUT.kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:10.1f} ms  avg {float(r['AverageNs'])/1e3:10.1f} us  {float(r['Percentage']):6.2f}%")
PY
cut -c1-300 $OUT.bench.json
