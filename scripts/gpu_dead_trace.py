#!/usr/bin/env python3
"""Durations of the dead-row kernels launch by launch (torch profiler): python scripts/gpu_dead_trace.py K B N iters hard"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from torch.profiler import ProfilerActivity, profile
from tclip_amd import engine, synth
K, B, N, iters, hard = (int(v) for v in sys.argv[1:6])
os.environ.setdefault("TCLIP_STREAM_GROUPS", "1")
x, _ = synth.make_query_tasks(B * N, K, seed=6); x = x.cuda()
run = lambda: engine.run_em_dirichlet(x, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=bool(hard))
run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    r = run(); torch.cuda.synchronize()
ev = sorted((e for e in prof.events() if e.device_type is not None and "tclip::k_mm" in e.name), key=lambda e: e.time_range.start)
seq = []
for e in ev:
    n = e.name
    tag = "H" if "k_mm_probe_head" in n else "P" if "k_mm_probe" in n else "D" if ("k_mm_live" in n and " true" in n) else "S" if "k_mm_split" in n else "L" if "k_mm_live" in n else None
    if tag: seq.append((tag, e.device_time))
tot = {}
for t, d in seq: tot[t] = tot.get(t, 0) + d
print("mm_iters", r.mm_iters[0].tolist())
print("totals us:", {k: round(v) for k, v in tot.items()})
# per outer iteration: the D launches in order
out, cur = [], []
for t, d in seq:
    if t in "DPH": cur.append(f"{t}{d:.0f}")
print(" ".join(cur[:400]))
