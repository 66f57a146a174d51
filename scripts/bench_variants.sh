#!/bin/bash
# usage: bash scripts/bench_variants.sh "<prof_small args>" <variant.so|orig> ... : times every variant on the given workload
cd $GRAFT_REPO_ROOT
args=$1; shift
cp transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
for v in "$@"; do
  [ "$v" != "orig" ] && cp $v transductive-clip_amd/tclip_amd/libtclip.so
  echo "== $v"
  timeout 600 python scripts/prof_small.py $args 2>&1 | tail -1 | cut -c1-150
  cp /tmp/libtclip_orig.so transductive-clip_amd/tclip_amd/libtclip.so
done
