#!/bin/bash
# usage (GPU box): bash scripts/gpu_pmc_ab.sh <label> "<prof_small args>" <lib> [<lib> ...]   (lib: a path, or `orig` for the tree's library)
# One PMC pass + kernel trace per library on the same workload; per MM kernel: VALU wave-instructions, their lane count per
# element-update, wave cycles, waits, summed durations -> gpurun_out/pmc_ab_<label>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
label=$1; wl=$2; shift 2
out=$R/gpurun_out/pmc_ab_$label.txt
: > $out
for lib in "$@"; do
  if [ "$lib" = orig ]; then export TCLIP_LIB=; else export TCLIP_LIB=$R/$lib; fi
  rm -rf $R/gpurun_out/pmcab
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcab -- python3 $R/scripts/prof_small.py $wl > $R/gpurun_out/pmcab.log 2>&1
  python3 - $R "$lib" "$wl" >> $out <<'PY'
import csv, sys, glob, collections, re
R, lib, wl = sys.argv[1:4]
f = glob.glob(f"{R}/gpurun_out/pmcab/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void tclip::", "").replace("tclip::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
        if "Start_Timestamp" in r: dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
log = open(f"{R}/gpurun_out/pmcab.log").read()
runs = re.findall(r"updates=([0-9.e+]+)", log)
upd = sum(float(u) for u in runs)
print(f"== {lib}  [{wl}]  element-updates {upd:.4e}  {[l for l in log.splitlines() if l.startswith('K=')][-1][:110]}")
tot = 0.0
for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"])[:6]:
    a = agg[k]
    if "k_mm_live" in k and "false" in k or "k_mm_split" in k: tot += a["SQ_INSTS_VALU"]
    print(f"  {k[:60]:60s} n={n[k]:5d} valu={a['SQ_INSTS_VALU']:.4e} salu={a['SQ_INSTS_SALU']:.3e} lds={a['SQ_INSTS_LDS']:.3e} wave_cycles={a['SQ_WAVE_CYCLES']:.4e} "
          f"wait={a['SQ_WAIT_ANY'] / max(a['SQ_WAVE_CYCLES'], 1):.3f} valu_active={a['SQ_ACTIVE_INST_VALU']:.4e} dur_ms={dur[k]:.1f}")
print(f"  lane-instructions per update (k_mm_live + k_mm_split): {tot * 64 / upd:.1f}")
PY
done
rm -rf $R/gpurun_out/pmcab
cat $out
