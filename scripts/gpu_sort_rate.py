#!/usr/bin/env python3
"""How often does k_mm_split sort its class queues anew?  (round 5: a wavefront keeps the placement of its elements across MM
iterations; tclip_profile_last_split_sorts counts wavefront-iterations and full placements)

    python scripts/gpu_sort_rate.py [K B N iters hard shots] ...
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from tclip_amd import engine, synth, _capi


def set_split(mode):
    _capi.check(_capi.lib().tclip_debug_set_mm_split(mode), 'tclip_debug_set_mm_split')

nums = [int(v) for v in sys.argv[1:]]
shapes = [tuple((nums[j:j + 6] + [0, 0])[:6]) for j in range(0, len(nums), 6)] or [(1000, 2, 125, 20, 0, 0), (100, 10, 100, 20, 0, 0), (397, 4, 100, 10, 1, 0), (1000, 2, 12, 6, 0, 1)]
for K, B, N, iters, hard, shots in shapes:
    x_q, _ = synth.make_query_tasks(B * N, K, seed=3, k_eff=(5 if shots else None))
    x_q = x_q.cuda()
    x_s = y_s = None
    if shots:
        x_s, y_s = synth.make_support(B * N, K, shots, seed=3)
        x_s, y_s = x_s.cuda(), y_s.squeeze(2).cuda()
    for mode in (-1, 1):            # default rule / k_mm_split from the first outer iteration on
        set_split(mode)
        engine.profile_enable(True)
        engine.profile_collect()
        torch.cuda.synchronize(); t = time.time()
        res = engine.run_em_dirichlet(x_q, x_s, y_s, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=bool(hard))
        torch.cuda.synchronize(); dt = time.time() - t
        busy, total, n, upd = engine.profile_collect()
        it, so = engine.profile_last_split_sorts()
        engine.profile_enable(False)
        print(f"K={K} B={B} N={N} iters={iters} hard={hard} shots={shots} split_mode={mode}: {dt:.3f}s  wave-iterations {it}  sorts {so} "
              f"({100.0 * so / max(it, 1):.2f} %)  mm_iters {res.mm_iters[0].tolist()}", flush=True)
    set_split(-1)
