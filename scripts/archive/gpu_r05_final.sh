#!/bin/bash
# round 5, last call: the gated suite, smoke(), the driver's bench command, the same headline through the driver's multi-GPU launcher with one rank
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05_final_suite.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a gpurun_out/r05_final_suite.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_bench.err | tail -1 > gpurun_out/r05_bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline 2> gpurun_out/r05_bench_torchrun.err | grep '^{' | tail -1 > gpurun_out/r05_bench_torchrun_1rank.json
cut -c1-300 gpurun_out/r05_bench.json; echo; cut -c1-300 gpurun_out/r05_bench_torchrun_1rank.json
