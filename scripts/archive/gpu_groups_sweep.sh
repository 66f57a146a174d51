#!/bin/bash
# stream-group sweep (TCLIP_STREAM_GROUPS x GPU_MAX_HW_QUEUES) on the K=100 and K=1000 bench shapes
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for q in 8 16; do
for g in 1 3 4 5 6 8 10; do
  echo "== groups $g queues $q: $(GPU_MAX_HW_QUEUES=$q TCLIP_STREAM_GROUPS=$g SPLIT_MODES=-1 timeout 600 python scripts/gpu_split_ab.py ${SHAPES:-100 10 100 20 1000 10 125 20} 2>&1 | grep '^K=' | cut -c1-60 | tr '\n' ' ')"
done
done
