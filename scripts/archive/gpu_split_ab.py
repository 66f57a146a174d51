#!/usr/bin/env python3
"""A/B timing of the class-split MM kernel (tclip_debug_set_mm_split 0 = never / -1 = default rule / 1 = always):
python scripts/gpu_split_ab.py [K B N iters]...  prints total seconds per mode and checks the results are identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import _capi, engine, synth

MODES = [int(m) for m in os.environ.get("SPLIT_MODES", "0,-1,1").split(",")]
shapes = [(1000, 2, 125, 6), (100, 10, 100, 20), (397, 4, 100, 10), (10, 10, 100, 20)]
if len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
for K, B, N, iters in shapes:
    x_q, _ = synth.make_query_tasks(B * N, K, seed=3)
    x_q = x_q.cuda()
    ref, line = None, []
    for mode in MODES:
        _capi.lib().tclip_debug_set_mm_split(mode)
        best = 1e9
        for rep in range(2):
            torch.cuda.synchronize(); t = time.time()
            res = engine.run_em_dirichlet(x_q, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=False)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        same = True if ref is None else all(torch.equal(getattr(res, n), getattr(ref, n)) for n in ("alpha", "u", "v", "mm_iters"))
        ref = res if ref is None else ref
        line.append(f"mode {mode:3d}: {best:.3f}s same={same}")
    _capi.lib().tclip_debug_set_mm_split(-1)
    print(f"K={K} B={B} N={N} iters={iters}  " + "  ".join(line) + f"  mm_iters={ref.mm_iters[0].tolist()}", flush=True)
