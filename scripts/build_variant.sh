#!/bin/bash
# usage: bash scripts/build_variant.sh <name> [-DFLAG=..]... : builds gpurun_variants/<name>.so from the working tree.
# Variants are selected at run time with TCLIP_LIB=<path> (tclip_amd/_capi.py); libtclip.so itself is never overwritten.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
HIPCC=${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}
mkdir -p "$ROOT/gpurun_variants"
cd "$ROOT/transductive-clip_amd/csrc" || exit 1
name=$1; shift
"$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -pthread -std=c++17 -Wall -Wno-unused-function "$@" -o "$ROOT/gpurun_variants/$name.so" tclip_kernels.hip tclip_host.cpp 2>&1 | grep -E "error|warning: v" | head
echo built $name
