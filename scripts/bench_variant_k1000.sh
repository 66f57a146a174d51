#!/bin/bash
# usage: bash scripts/bench_variant_k1000.sh <variant.so> ... : like bench_variant.sh on the K=1000 shape (2 batches x 125 tasks, 3 outer iterations)
cd $GRAFT_REPO_ROOT
cp transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
for v in "$@"; do
  [ "$v" != "orig" ] && cp $v transductive-clip_amd/tclip_amd/libtclip.so
  echo "== $v"
  timeout 300 python scripts/prof_small.py 1000 2 125 3 0 2>&1 | tail -1 | cut -c1-120
  cp /tmp/libtclip_orig.so transductive-clip_amd/tclip_amd/libtclip.so
done
