#!/usr/bin/env python3
"""Per-kernel device time of one EM-Dirichlet engine call on ONE stream (torch profiler): python scripts/gpu_kernel_totals.py K B N iters hard [shots]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
os.environ.setdefault("TCLIP_STREAM_GROUPS", "1")
import torch
from torch.profiler import ProfilerActivity, profile
from tclip_amd import engine, synth
K, B, N, iters, hard = (int(v) for v in sys.argv[1:6])
shots = int(sys.argv[6]) if len(sys.argv) > 6 else 0
x, _ = synth.make_query_tasks(B * N, K, seed=6, k_eff=(5 if shots else None)); x = x.cuda()
xs = ys = None
if shots:
    xs, ys = synth.make_support(B * N, K, shots, seed=6); xs, ys = xs.cuda(), ys.squeeze(2).cuda()
run = lambda: engine.run_em_dirichlet(x, xs, ys, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=bool(hard))
run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    r = run(); torch.cuda.synchronize()
rows = sorted(((e.device_time_total, e.count, e.key) for e in prof.key_averages() if e.device_time_total > 0), reverse=True)
tot = sum(t for t, _, _ in rows)
print(f"K={K} B={B} N={N} iters={iters} hard={hard} shots={shots}: {tot / 1e3:.1f} ms of kernels, mm_iters {r.mm_iters[0].tolist()}")
for t, c, k in rows[:14]:
    print(f"  {k.replace('void tclip::', '').replace('tclip::', '')[:58]:58s} calls {c:5d}  {t / 1e3:9.2f} ms  {100 * t / tot:5.2f} %")
