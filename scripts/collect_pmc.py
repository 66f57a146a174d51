#!/usr/bin/env python3
"""Merges gpurun_out/pmc_k1000.json / pmc_k100.json (scripts/gpu_pmc_json.sh) into profiles/pmc_current.json, the file
bench.py reads for roofline.traffic, and keeps per-round copies under profiles/."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
from tclip_amd import _capi  # noqa: E402
digest = _capi.source_digest()          # the kernel sources the counters were taken on (bench.py nulls stale PMC fields)
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {}
for key in ("k1000", "k100", "k397_hard", "fs_k1000"):
    p = os.path.join(ROOT, "gpurun_out", f"pmc_{key}.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    json.dump(d, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{key}.json"), "w"), indent=1)
    s = d["summary"]
    s["file"] = f"profiles/{tag}_pmc_{key}.json"
    s["commit"] = commit
    s["csrc_sha1"] = digest
    out[key] = s
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_current.json"), "w"), indent=1)
print(json.dumps({k: {f: v[f] for f in ("lane_instr_per_update", "wait_frac", "traffic_bytes_per_launch", "algorithmic_bytes_per_launch")} for k, v in out.items()}, indent=1))
