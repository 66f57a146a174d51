#!/bin/bash
# PMC passes of the SOFT_KMEANS kernels (k_kmeans_logits_tile, k_mstats_cols75) on 1000 tasks at K = 397 -> gpurun_out/pmc_kmeans.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_kmeans.txt
: > $out
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_[A-Z_0-9]*\|SQ_INSTS_SMEM[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*" | sort -u | tr '\n' ' ' >> $out; echo >> $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS"; do
  rm -rf $R/gpurun_out/pmck
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmck -- python3 $R/scripts/prof_kmeans.py 397 1000 20 > $R/gpurun_out/pmck.log 2>&1
  python3 - $R "$set" >> $out <<'PY'
import csv, sys, glob, collections
R, sets = sys.argv[1:3]
fs = glob.glob(f"{R}/gpurun_out/pmck/**/*counter_collection.csv", recursive=True)
if not fs:
    print("pass failed:", sets, open(f"{R}/gpurun_out/pmck.log").read()[-400:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void tclip::", "").replace("tclip::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
        if "Start_Timestamp" in r: dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
print("== pass:", sets)
for k in sorted(agg, key=lambda k: -dur[k])[:4]:
    print(f"  {k[:40]:40s} n={n[k]:4d} dur_ms={dur[k]:8.2f}  " + "  ".join(f"{c}={v:.4e}" for c, v in sorted(agg[k].items())))
PY
  i=$((i+1))
done
rm -rf $R/gpurun_out/pmck
cat $out
