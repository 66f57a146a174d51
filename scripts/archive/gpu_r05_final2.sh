#!/bin/bash
# round 5, after the last kernel-source change (the keep_placement test hook): PMC passes again (pmc_current.json is keyed by the
# source digest), the gated suite, the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KEY=k1000 bash scripts/gpu_pmc_json.sh 1000 4 125 20 > gpurun_out/pmc_k1000.log 2>&1
KEY=k100 bash scripts/gpu_pmc_json.sh 100 10 100 20 > gpurun_out/pmc_k100.log 2>&1
KEY=k397_hard bash scripts/gpu_pmc_json.sh 397 4 100 10 1 > gpurun_out/pmc_k397_hard.log 2>&1
KEY=fs_k1000 bash scripts/gpu_pmc_json.sh 1000 4 25 20 0 4 > gpurun_out/pmc_fs_k1000.log 2>&1
rm -f gpurun_out/pmc_*.trace*.csv
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05_final_suite.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a gpurun_out/r05_final_suite.txt
