#!/bin/bash
# usage (on the GPU box): bash scripts/gpu_quick.sh
cd $GRAFT_REPO_ROOT
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 400 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_first.json
