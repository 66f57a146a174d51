#!/bin/bash
# gated GPU tests, then the K=1000 bench shape timing (prof_small) and optionally the bench itself
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python scripts/prof_small.py 1000 10 125 20 2>&1 | tail -2 | cut -c1-200
timeout 120 python scripts/prof_small.py 100 10 100 20 2>&1 | tail -1 | cut -c1-200
if [ "$1" == "bench" ]; then timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_check.json; cut -c1-1500 gpurun_out/bench_check.json; fi
