#!/bin/bash
# usage (GPU box): bash scripts/gpu_ab_only.sh <label> <lib> [<lib> ...]: sort rates of the tree's library, then scripts/gpu_ab_libs.py on the given libraries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
label=$1; shift
timeout 900 python scripts/gpu_ab_libs.py "$@" 2>&1 | tee gpurun_out/r05_ab_$label.txt
