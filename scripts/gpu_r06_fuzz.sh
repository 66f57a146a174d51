#!/bin/bash
# round 6: randomised parity sweeps of the final MM kernels against the C++ oracle (scripts/gpu_fuzz.py), every case twice
# usage: bash scripts/gpu_r06_fuzz.sh [seed base, default 6000] [scale: 1 = the committed set, 2 = twice as many]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; base=${1:-7000}; scale=${2:-1}; out=gpurun_out/r06_fuzz_$base.txt; : > $out
run() { echo "\$ $*" >> $out; timeout 2400 "$@" 2>&1 | grep -v amdgpu.ids | tail -2 >> $out; }
run python scripts/gpu_fuzz.py $((600 * scale)) $((base + 1))
run python scripts/gpu_fuzz.py $((100 * scale)) $((base + 2)) large
run python scripts/gpu_fuzz.py $((80 * scale)) $((base + 3)) full
TCLIP_FUZZ_ROWSET_MIN_ROWS=0 run python scripts/gpu_fuzz.py $((150 * scale)) $((base + 4))
cat $out
