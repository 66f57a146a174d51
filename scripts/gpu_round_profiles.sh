#!/bin/bash
# A round's measurement set (GPU box): bash scripts/gpu_round_profiles.sh <tag, e.g. r06>
# PMC passes of the MM kernels on the workloads the bench runs - the K = 1000 one over the FULL 20 x 1000 schedule (what the
# headline executes), the few-shot one with 100 tasks -, kernel stats of every bench workload, one-stream stats of the K = 1000
# and few-shot shapes, sort rates of k_mm_split, counters of the SOFT_KMEANS kernels, then the driver's bench command.
# Results land in gpurun_out/; scripts/collect_round.py <tag> (build container) copies the summaries into profiles/.
cd $GRAFT_REPO_ROOT
tag=${1:-r06}
mkdir -p gpurun_out
KEY=k1000 bash scripts/gpu_pmc_json.sh 1000 4 125 20 > gpurun_out/pmc_k1000.log 2>&1
KEY=k100 bash scripts/gpu_pmc_json.sh 100 10 100 20 > gpurun_out/pmc_k100.log 2>&1
KEY=k397_hard bash scripts/gpu_pmc_json.sh 397 4 100 10 1 > gpurun_out/pmc_k397_hard.log 2>&1
KEY=fs_k1000 bash scripts/gpu_pmc_json.sh 1000 4 25 20 0 4 > gpurun_out/pmc_fs_k1000.log 2>&1
for w in k1000 k100 k397_hard fs_k1000; do bash scripts/gpu_prof_bench.sh $w > gpurun_out/prof_bench_$w.log 2>&1; done
TCLIP_STREAM_GROUPS=1 TAG=${tag}_single_stream bash scripts/gpu_kernel_stats.sh 1000 10 125 20 > gpurun_out/prof_${tag}_single_stream.log 2>&1
TCLIP_STREAM_GROUPS=1 TAG=${tag}_single_stream_fs bash scripts/gpu_kernel_stats.sh 1000 4 25 20 0 4 > gpurun_out/prof_${tag}_single_stream_fs.log 2>&1
rm -f gpurun_out/prof_${tag}_single_stream*.kernel_trace_full.csv gpurun_out/pmc_*.trace*.csv gpurun_out/pmc_*.pass*.csv
timeout 300 python scripts/gpu_sort_rate.py 1000 4 125 20 0 0 100 10 100 20 0 0 397 4 100 10 1 0 1000 4 25 20 0 4 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_split_sort_rate.txt
bash scripts/gpu_pmc_kmeans.sh > /dev/null 2>&1; cp gpurun_out/pmc_kmeans.txt gpurun_out/${tag}_pmc_kmeans.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json
cp gpurun_out/bench_full.json gpurun_out/${tag}_bench_full.json
ls -la gpurun_out | tail -30
cut -c1-600 gpurun_out/${tag}_bench.json
