#!/bin/bash
# usage: bash scripts/bench_variant_shapes.sh <variant.so> ... : K=1000 / K=397 / K=100 full-schedule wall times per variant
cd $GRAFT_REPO_ROOT
cp transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
for v in "$@"; do
  [ "$v" != "orig" ] && cp $v transductive-clip_amd/tclip_amd/libtclip.so
  echo "== $v"
  timeout 600 python scripts/gpu_shapes.py 2>&1 | grep "^K=1000 zero\|^K=397\|^K=10 " | cut -c1-70
  timeout 300 python scripts/prof_small.py 100 10 100 20 0 2>&1 | tail -1 | cut -c1-60
  cp /tmp/libtclip_orig.so transductive-clip_amd/tclip_amd/libtclip.so
done
