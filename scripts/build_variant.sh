#!/bin/bash
# usage: bash scripts/build_variant.sh <name> [-DFLAG=..]... : builds gpurun_variants/<name>.so from the working tree
cd /root/repo/transductive-clip_amd/csrc
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -pthread -std=c++17 -Wall -Wno-unused-function "$@" -o /root/repo/gpurun_variants/$name.so tclip_kernels.hip tclip_host.cpp 2>&1 | grep -E "error|warning: v" | head
echo built $name
