#!/bin/bash
# round 5 (GPU box): the gated suite, then the literal configs[3] job on ONE GPU (80 batches of 125 K = 1000 tasks = 10 000 tasks,
# --scaling strong): the N = 1 anchor of a strong-scaling curve -> gpurun_out/r05_bench_strong_1gpu.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r05_gpu_suite.txt
timeout 900 python bench.py --gpus 1 --scaling strong --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> gpurun_out/r05_bench_strong_1gpu.err | tail -1 > gpurun_out/r05_bench_strong_1gpu.json
tail -3 gpurun_out/r05_bench_strong_1gpu.err
cut -c1-600 gpurun_out/r05_bench_strong_1gpu.json
