#!/bin/bash
# time of k_logits (summed, one stream) for each variant library: bash scripts/gpu_logits_time.sh <variant.so|orig> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
for v in "$@"; do
  [ "$v" != "orig" ] && cp $R/$v $R/transductive-clip_amd/tclip_amd/libtclip.so
  OUT=$R/gpurun_out/prof_lg
  TCLIP_STREAM_GROUPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/prof_small.py 1000 4 125 2 > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep K= $OUT.log | tail -1 | cut -c1-50) $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_logits' in r['Name']: print('k_logits calls', r['Calls'], 'total ms', round(float(r['TotalDurationNs'])/1e6,1))")"
  rm -rf $OUT
  cp /tmp/libtclip_orig.so $R/transductive-clip_amd/tclip_amd/libtclip.so
done
