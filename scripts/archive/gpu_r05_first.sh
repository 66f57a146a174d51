#!/bin/bash
# round 5, first GPU call: the lazy placement of k_mm_split - parity suite, then sort rates, then the A/B against the same tree
# compiled with TCLIP_SPLIT_LAZY=0 (sorts every iteration, as round 4 did)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_parity_golden.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r05_first_tests.txt
timeout 600 python scripts/gpu_sort_rate.py 2>&1 | tee gpurun_out/r05_sort_rate.txt
timeout 900 python scripts/gpu_ab_libs.py gpurun_variants/nolazy.so orig 2>&1 | tee gpurun_out/r05_ab_lazy.txt
