"""SOFT_KMEANS workload for kernel-time breakdown: python3 scripts/prof_kmeans.py [K] [tasks] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 397
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
x, _ = synth.make_query_tasks(T, K, seed=6); x = x.cuda()
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    engine.run_soft_kmeans(x, iters=iters, temperature=30)
    torch.cuda.synchronize()
    print(f"SOFT_KMEANS K={K} T={T} iters={iters}: {(time.time() - t) * 1e3:.1f} ms", flush=True)
