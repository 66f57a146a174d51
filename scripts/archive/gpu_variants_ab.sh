#!/bin/bash
# usage (GPU box): bash scripts/gpu_variants_ab.sh "<gpu_split_ab args>" orig gpurun_variants/a.so ... : the A/B timing of
# scripts/gpu_split_ab.py for the installed library and each variant (loaded through TCLIP_LIB)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
args=$1; shift
for v in "$@"; do
  if [ "$v" = "orig" ]; then lib=""; else lib=$(realpath $v); fi
  echo "== $v"
  TCLIP_LIB=$lib SPLIT_MODES=${SPLIT_MODES:-0,-1} timeout 900 python scripts/gpu_split_ab.py $args 2>&1 | grep "^K=" | cut -c1-140
done
