#!/bin/bash
# counter profile of a small fixed workload (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${PMC_TAG:-r01}
DEFAULT_SETS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD|GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE|FETCH_SIZE|WRITE_SIZE"
IFS='|' read -ra SETS <<< "${PMC_SETS:-$DEFAULT_SETS}"
for set in "${SETS[@]}"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/scripts/prof_small.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
  f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$R/gpurun_out/pmc_${TAG}_$tag.summary.txt" <<'PY'
import csv, sys, collections
f, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(r["Dispatch_Id"],k)
    if key not in seen: seen.add(key); cnt[k]+=1
with open(out, "w") as o:
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:6]:
        o.write(f"{k} dispatches={cnt[k]} " + " ".join(f"{c}={v:.5g}" for c, v in agg[k].items()) + "\n")
print(open(out).read())
PY
  rm -rf $R/gpurun_out/pmc_$tag
done
grep "K=" $R/gpurun_out/pmc_SQ_WAVES.log | tee $R/gpurun_out/pmc_${TAG}_workload.txt
