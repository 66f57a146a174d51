#!/bin/bash
# usage: bash scripts/bench_variant.sh <variant.so> : temporarily swaps the library and runs the bench
cd $GRAFT_REPO_ROOT
cp transductive-clip_amd/tclip_amd/libtclip.so /tmp/libtclip_orig.so
for v in "$@"; do
  cp $v transductive-clip_amd/tclip_amd/libtclip.so
  echo "== $v"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_busy_ms_per_step'])"
done
cp /tmp/libtclip_orig.so transductive-clip_amd/tclip_amd/libtclip.so
