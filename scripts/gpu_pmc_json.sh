#!/bin/bash
# PMC passes (separate --pmc runs, kernel-trace only) of scripts/prof_small.py "$@", summarised for the dominant kernel
# into gpurun_out/pmc_$KEY.json (KEY env: k1000 / k100).  FETCH_SIZE / WRITE_SIZE are calibrated on k_copy of the same
# run: a dword-per-lane grid-stride copy of a known byte count, the MM kernel's own access shape.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KEY=${KEY:-k1000}
SETS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD|GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE|FETCH_SIZE|WRITE_SIZE"
IFS='|' read -ra PASSES <<< "$SETS"
rm -f $R/gpurun_out/pmc_$KEY.*.csv
i=0
for set in "${PASSES[@]}"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcdir_$i -- python3 $R/scripts/prof_small.py "$@" > $R/gpurun_out/pmc_$KEY.pass$i.log 2>&1
  f=$(find $R/gpurun_out/pmcdir_$i -name "*counter_collection.csv" | head -1)
  cp $f $R/gpurun_out/pmc_$KEY.pass$i.csv
  t=$(find $R/gpurun_out/pmcdir_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && cp $t $R/gpurun_out/pmc_$KEY.trace$i.csv
  rm -rf $R/gpurun_out/pmcdir_$i
  i=$((i+1))
done
python3 - $R $KEY "$*" <<'PY'
import csv, sys, json, glob, collections, re
R, key, wl = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
extra = {}
for f in sorted(glob.glob(f"{R}/gpurun_out/pmc_{key}.pass*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void tclip::", "").replace("tclip::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
        if ("k_mm_live" in k and "false" in k) or "k_mm_split" in k:
            extra = {"scratch_bytes_per_lane": int(r.get("Scratch_Size", 0) or 0), "vgpr": int(r.get("VGPR_Count", 0) or 0),
                     "lds_bytes_per_block": int(r.get("LDS_Block_Size", 0) or 0)}
# the clock the chip held under the MM kernels: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time, both from the
# pass that counted GRBM_GUI_ACTIVE (kernel time from that pass's kernel trace, joined on the dispatch id)
clock_ghz = None
try:
    gui = collections.defaultdict(float)
    for f in sorted(glob.glob(f"{R}/gpurun_out/pmc_{key}.pass*.csv")):
        rows = list(csv.DictReader(open(f)))
        if not any(r["Counter_Name"] == "GRBM_GUI_ACTIVE" for r in rows):
            continue
        i = re.search(r"pass(\d+)", f).group(1)
        for r in rows:
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and (("k_mm_live" in r["Kernel_Name"] and "false" in r["Kernel_Name"]) or "k_mm_split" in r["Kernel_Name"]):
                gui[r["Dispatch_Id"]] += float(r["Counter_Value"])
        dur = {}
        if rows and "Start_Timestamp" in rows[0] and "End_Timestamp" in rows[0]:
            for r in rows:
                dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        else:
            for r in csv.DictReader(open(f"{R}/gpurun_out/pmc_{key}.trace{i}.csv")):
                dur[r.get("Dispatch_Id", r.get("Correlation_Id"))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        # long dispatches only: the quotient reads high on dispatches shorter than ~0.3 ms (guide, DVFS section)
        ids = [d for d in gui if d in dur and dur[d] > 3e5]
        if ids:
            clock_ghz = sum(gui[d] for d in ids) / 8.0 / sum(dur[d] for d in ids)
except Exception as e:
    print("clock not derived:", e)
log = open(f"{R}/gpurun_out/pmc_{key}.pass0.log").read()
runs = re.findall(r"K=(\d+) B=(\d+) N=(\d+) iters=(\d+) .*?launches=(\d+) updates=([0-9.e+]+)", log)
K, B, N, iters, launches, updates = runs[-1]
K, B, N = int(K), int(B), int(N)
n_runs = len(runs)
updates_total = sum(float(r[5]) for r in runs)
# the live rows run through k_mm_live<.., false, ..> in the first outer iteration and through k_mm_split afterwards: the
# summary covers both (the element-update counter does), `per_kernel` keeps them apart
mm = [k for k in agg if ("k_mm_live" in k and "false" in k) or "k_mm_split" in k]
live = " + ".join(sorted(mm))
a = collections.defaultdict(float)
for k in mm:
    for c, v in agg[k].items():
        a[c] += v
n_disp = sum(len(cnt[(k, "SQ_INSTS_VALU")]) for k in mm)
copy_disp = len(cnt[("k_copy", "FETCH_SIZE")])
copy_bytes = 4.0 * B * N * 75 * K * n_runs                       # every run copies x_q -> u once (split over the stream groups)
fetch_ratio = agg["k_copy"]["FETCH_SIZE"] * 1024 / copy_bytes if copy_bytes else None
write_ratio = agg["k_copy"]["WRITE_SIZE"] * 1024 / copy_bytes if copy_bytes else None
fetch = a["FETCH_SIZE"] * 1024 / n_disp
write = a["WRITE_SIZE"] * 1024 / n_disp
out = {
    "workload": f"scripts/prof_small.py {wl} (K={K}, {B} batches x {N} tasks, {iters} outer iterations, run {n_runs}x)",
    "kernel": live, "dispatches": n_disp,
    "lane_instr_per_update": a["SQ_INSTS_VALU"] * 64 / updates_total,
    "valu_wave_instr": a["SQ_INSTS_VALU"], "element_updates": updates_total,
    "per_kernel": {k: {"dispatches": len(cnt[(k, "SQ_INSTS_VALU")]), "valu_wave_instr": agg[k]["SQ_INSTS_VALU"],
                       "wave_cycles": agg[k]["SQ_WAVE_CYCLES"], "wait_frac": agg[k]["SQ_WAIT_ANY"] / max(agg[k]["SQ_WAVE_CYCLES"], 1),
                       "valu_active_quad_cycles": agg[k]["SQ_ACTIVE_INST_VALU"], "gui_active": agg[k]["GRBM_GUI_ACTIVE"],
                       "valu_busy_frac": agg[k]["SQ_ACTIVE_INST_VALU"] * 4 / max(agg[k]["GRBM_GUI_ACTIVE"] / 8 * 1024, 1)} for k in mm},
    "valu_busy_frac": a["SQ_ACTIVE_INST_VALU"] * 4 / max(a["GRBM_GUI_ACTIVE"] / 8 * 1024, 1),
    "clock_ghz": clock_ghz,
    "wait_frac": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], "issue_stall_frac": a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
    "active_frac": a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"],
    "trans_frac_of_valu": a["SQ_INSTS_VALU_TRANS_F32"] / a["SQ_INSTS_VALU"],
    "lds_conflict_frac": a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1),
    "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch_raw": write,
    "calibration": {"kernel": "k_copy (dword per lane, grid-stride)", "known_bytes_each_way": copy_bytes,
                    "fetch_counter_over_bytes": fetch_ratio, "write_counter_over_bytes": write_ratio},
    "fetch_bytes_per_launch": fetch / fetch_ratio if fetch_ratio else None,
    "write_bytes_per_launch": write / write_ratio if write_ratio else None,
    "algorithmic_bytes_per_launch": 12.0 * updates_total / 50.0 / n_disp * (n_disp / max(n_disp, 1)),
    **extra,
}
out["traffic_bytes_per_launch"] = (out["fetch_bytes_per_launch"] or 0) + (out["write_bytes_per_launch"] or 0)
others = {k: {c: v for c, v in agg[k].items()} for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:6]}
json.dump({"summary": out, "counters_by_kernel": others}, open(f"{R}/gpurun_out/pmc_{key}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -f $R/gpurun_out/pmc_$KEY.pass*.csv $R/gpurun_out/pmc_$KEY.trace*.csv
