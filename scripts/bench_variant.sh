#!/bin/bash
# usage: bash scripts/bench_variant.sh <variant.so> ... : swaps the library in turn and times the
# K=100 workload with scripts/prof_small.py (no accuracy tail, so experimental builds that break
# the numerics can still be timed); the original library is restored afterwards.
cd $GRAFT_REPO_ROOT
cp transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
for v in "$@"; do
  [ "$v" != "orig" ] && cp $v transductive-clip_amd/tclip_amd/libtclip.so
  echo "== $v"
  timeout 300 python scripts/prof_small.py 100 10 100 20 0 2>&1 | tail -1 | cut -c1-120
  cp /tmp/libtclip_orig.so transductive-clip_amd/tclip_amd/libtclip.so
done
