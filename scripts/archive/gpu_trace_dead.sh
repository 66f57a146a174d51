#!/bin/bash
# kernel trace of a short run, k_mm_chunk / k_mm_live durations in dispatch order per stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TCLIP_STREAM_GROUPS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_dead -- python3 $R/scripts/prof_small.py 100 10 100 4 > $R/gpurun_out/trace_dead.log 2>&1
f=$(find $R/gpurun_out/trace_dead -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/trace_dead_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "k_mm_chunk" in n or "k_mm_live" in n:
        seq.append(("D" if "k_mm_chunk" in n else "L", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# second repetition only (the script runs the workload twice)
half = len(seq) // 2
out = []
for kind, us in seq[half:]:
    out.append(f"{kind}{us:.0f}")
print(" ".join(out))
# idle gaps between consecutive kernels of the (single) stream, steady state: last 60 kernels
allk = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-14:]) for r in rows]
tail = allk[-60:]
print("gaps_us:", " ".join(f"{tail[i][2]}>{(tail[i + 1][0] - tail[i][1]) / 1e3:.1f}" for i in range(len(tail) - 1)))
PY
rm -rf $R/gpurun_out/trace_dead
