#!/bin/bash
# usage: bash scripts/isa_one.sh <out-prefix> [-DFLAG=..]... : the K = 1000 MM kernels alone (TCLIP_ISA_ONLY) to <out-prefix>.s with the
# compiler's resource-usage remarks in <out-prefix>.txt - seconds per compile; then e.g.
#   python scripts/isa_blocks.py <out-prefix>.s _ZN5tclip10k_mm_splitILi16ELi64ELi1000EEEvNS_6MMArgsE 20
ROOT=$(cd "$(dirname "$0")/.." && pwd)
out=$1; shift
cd "$ROOT/transductive-clip_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Wno-unused-function --cuda-device-only -S \
  -DTCLIP_ISA_ONLY -Rpass-analysis=kernel-resource-usage "$@" -o "$out.s" tclip_kernels.hip 2> "$out.txt"
grep -E "Function Name|VGPRs:|ScratchSize|Spill|Occupancy" "$out.txt" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
