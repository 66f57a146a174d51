#!/bin/bash
# Round-end artefacts (on the GPU box): bash scripts/gpu_final.sh ; then copy gpurun_out/final_* into profiles/
cd $GRAFT_REPO_ROOT
timeout 500 python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/final_bench.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_torchrun.json
bash scripts/gpu_prof.sh final > /dev/null 2>&1
python3 scripts/gpu_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final_shapes.txt
cut -c1-140 gpurun_out/final_bench.json gpurun_out/final_bench_torchrun.json
head -4 gpurun_out/prof_final.kernel_stats.csv | cut -c1-150
