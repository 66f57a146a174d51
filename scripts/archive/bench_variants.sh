#!/bin/bash
# usage: bash scripts/bench_variants.sh "<prof_small args>" <variant.so|orig> ... : times every variant on the given workload
# (the variant is loaded through TCLIP_LIB; the installed libtclip.so is not touched)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
args=$1; shift
for v in "$@"; do
  echo "== $v"
  if [ "$v" = "orig" ]; then lib=""; else lib=$(realpath $v); fi
  TCLIP_LIB=$lib timeout 600 python scripts/prof_small.py $args 2>&1 | tail -1 | cut -c1-150
done
