cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/r05_fuzz.txt; : > $out
run() { echo "\$ $*" >> $out; timeout 1500 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> $out; }
run python scripts/gpu_fuzz.py 600 6001
run python scripts/gpu_fuzz.py 100 6002 large
run python scripts/gpu_fuzz.py 80 6003 full
TCLIP_FUZZ_ROWSET_MIN_ROWS=0 run python scripts/gpu_fuzz.py 150 6004
cat $out
