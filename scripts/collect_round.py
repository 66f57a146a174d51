#!/usr/bin/env python3
"""Build container, after scripts/gpu_round_profiles.sh <tag> ran on the GPU box: copies the round's summaries from gpurun_out/ into
profiles/ (tracked) and refreshes profiles/pmc_current.json (scripts/collect_pmc.py).  python scripts/collect_round.py r06"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [(f"prof_bench_{w}.kernel_stats.csv", f"{tag}_kernel_stats_{w}.csv") for w in ("k1000", "k100", "k397_hard", "fs_k1000")]
pairs += [(f"prof_bench_{w}.kernel_trace_head.csv", f"{tag}_kernel_trace_head_{w}.csv") for w in ("k1000", "k100", "k397_hard", "fs_k1000")]
pairs += [(f"prof_{tag}_single_stream.kernel_stats.csv", f"{tag}_kernel_stats_k1000_single_stream.csv"),
          (f"prof_{tag}_single_stream_fs.kernel_stats.csv", f"{tag}_kernel_stats_fs_k1000_single_stream.csv"),
          (f"{tag}_split_sort_rate.txt", f"{tag}_split_sort_rate.txt"), (f"{tag}_pmc_kmeans.txt", f"{tag}_pmc_kmeans.txt"),
          (f"{tag}_bench.json", f"{tag}_bench_line.json"), (f"{tag}_bench_full.json", f"{tag}_bench.json")]
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "collect_pmc.py"), tag])
