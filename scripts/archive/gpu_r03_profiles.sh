#!/bin/bash
# round-3 measurement set (GPU box): PMC passes of the MM kernels, kernel stats of every bench workload, a short bench run
cd $GRAFT_REPO_ROOT
KEY=k1000 bash scripts/gpu_pmc_json.sh 1000 2 125 6 > gpurun_out/pmc_k1000.log 2>&1
KEY=k100 bash scripts/gpu_pmc_json.sh 100 10 100 20 > gpurun_out/pmc_k100.log 2>&1
KEY=k397_hard bash scripts/gpu_pmc_json.sh 397 4 100 10 1 > gpurun_out/pmc_k397_hard.log 2>&1
KEY=fs_k1000 bash scripts/gpu_pmc_json.sh 1000 2 12 6 0 4 > gpurun_out/pmc_fs_k1000.log 2>&1
for w in k1000 k100 k397_hard fs_k1000; do bash scripts/gpu_prof_bench.sh $w > gpurun_out/prof_bench_$w.log 2>&1; done
TCLIP_STREAM_GROUPS=1 TAG=r03_single_stream bash scripts/gpu_r02_stats.sh 1000 10 125 20 > gpurun_out/prof_r03_single_stream.log 2>&1
(time python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err) 2>&1 | tail -3
tail -c 300 gpurun_out/r03_bench.json
