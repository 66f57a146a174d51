#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "kmeans" 2>&1 | tail -3
timeout 300 python scripts/gpu_kmeans_live.py 397 1000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_kmeans_live.txt
timeout 600 python bench.py --workload k397_hard --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_k397.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_k397.json"))
s = d.get("secondary", {}).get("k397_hard", d)
print({k: s.get(k) for k in ("value", "ms_per_step")}, s.get("soft_kmeans", {}).get("value"), s.get("soft_kmeans", {}).get("ms_per_step"))
PY
