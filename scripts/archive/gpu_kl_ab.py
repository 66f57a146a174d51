#!/usr/bin/env python3
"""KL_KMEANS engine time with the round-3 kernels (tclip_debug_set_kmeans_tile 0) and the default rule: python scripts/gpu_kl_ab.py [K T iters]..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from tclip_amd import _capi, engine, synth
shapes = [(397, 1000, 10), (100, 1000, 10)]
if len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for K, T, iters in shapes:
    x, _ = synth.make_query_tasks(T, K, seed=6); x = x.cuda()
    line, ref = [], None
    for mode in (0, -1):
        _capi.lib().tclip_debug_set_kmeans_tile(mode)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.time()
            res = engine.run_kl_kmeans(x, iters=iters)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        same = ref is None or all(torch.equal(a, b) for a, b in zip(res, ref))
        ref = ref or res
        line.append(f"mode {mode}: {best * 1e3:.1f} ms same={same}")
    _capi.lib().tclip_debug_set_kmeans_tile(-1)
    print(f"KL_KMEANS K={K} T={T} iters={iters}  " + "  ".join(line), flush=True)
