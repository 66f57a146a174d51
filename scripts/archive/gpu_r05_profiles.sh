#!/bin/bash
# round-5 measurement set (GPU box): PMC passes of the MM kernels on the workloads the bench runs - the K = 1000 one over the
# FULL 20 x 1000 schedule (what the headline executes), the few-shot one with 100 tasks -, kernel stats of every bench workload,
# one-stream stats of the K = 1000 and few-shot shapes, sort rates of k_mm_split, then the driver's bench command.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KEY=k1000 bash scripts/gpu_pmc_json.sh 1000 4 125 20 > gpurun_out/pmc_k1000.log 2>&1
KEY=k100 bash scripts/gpu_pmc_json.sh 100 10 100 20 > gpurun_out/pmc_k100.log 2>&1
KEY=k397_hard bash scripts/gpu_pmc_json.sh 397 4 100 10 1 > gpurun_out/pmc_k397_hard.log 2>&1
KEY=fs_k1000 bash scripts/gpu_pmc_json.sh 1000 4 25 20 0 4 > gpurun_out/pmc_fs_k1000.log 2>&1
for w in k1000 k100 k397_hard fs_k1000; do bash scripts/gpu_prof_bench.sh $w > gpurun_out/prof_bench_$w.log 2>&1; done
TCLIP_STREAM_GROUPS=1 TAG=r05_single_stream bash scripts/gpu_kernel_stats.sh 1000 10 125 20 > gpurun_out/prof_r05_single_stream.log 2>&1
TCLIP_STREAM_GROUPS=1 TAG=r05_single_stream_fs bash scripts/gpu_kernel_stats.sh 1000 4 25 20 0 4 > gpurun_out/prof_r05_single_stream_fs.log 2>&1
rm -f gpurun_out/prof_r05_single_stream*.kernel_trace_full.csv gpurun_out/pmc_*.trace*.csv
timeout 300 python scripts/gpu_sort_rate.py 1000 4 125 20 0 0 100 10 100 20 0 0 397 4 100 10 1 0 1000 4 25 20 0 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_split_sort_rate.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_bench.err | tail -1 > gpurun_out/r05_bench.json
ls -la gpurun_out | tail -30
cut -c1-400 gpurun_out/r05_bench.json
