#!/bin/bash
# usage: bash scripts/build_rev_variant.sh <name> <git-rev> [-DFLAG=..]... : builds gpurun_variants/<name>.so from the kernel sources
# of an earlier commit (csrc/ and include/ as they were at <git-rev>), for A/B timing against the working tree
# (scripts/gpu_ab_libs.py); nothing in the working tree is touched.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}
name=$1; rev=$2; shift 2
tmp=$(mktemp -d)
mkdir -p "$ROOT/gpurun_variants"
git -C "$ROOT" archive "$rev" transductive-clip_amd/csrc include | tar -x -C "$tmp" || exit 1
# the host half and the header of the working tree: entry points added since <git-rev> must exist for the binding to load
if [ -n "$HOST_FROM_TREE" ]; then cp "$ROOT/transductive-clip_amd/csrc/tclip_host.cpp" "$tmp/transductive-clip_amd/csrc/"; cp "$ROOT/include/tclip.h" "$tmp/include/"; fi
cd "$tmp/transductive-clip_amd/csrc" || exit 1
"$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -pthread -std=c++17 -Wall -Wno-unused-function "$@" -o "$ROOT/gpurun_variants/$name.so" tclip_kernels.hip tclip_host.cpp 2>&1 | grep -E "error" | head
rm -rf "$tmp"
echo built $name from $rev
