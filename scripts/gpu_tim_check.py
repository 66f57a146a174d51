#!/usr/bin/env python3
"""ALPHA_TIM and LAPLACIAN_SHOT on the GPU: deviations from the reference fixtures and timings."""
import glob
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
from tclip_amd import engine, synth  # noqa: E402

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fs_tim_*.npz"))):
    g = np.load(path)
    w, lq, preds, crit = engine.run_alpha_tim(
        torch.from_numpy(g["x_q"]).cuda(), torch.from_numpy(g["x_s"]).cuda(), torch.from_numpy(g["y_s"]).squeeze(2).cuda(),
        iters=int(g["iters"]), temp=float(g["temp"]), lr=float(g["lr"]), alpha_value=float(g["alpha_value"]),
        loss_weights=[float(x) for x in g["loss_weights"]], entropies=[str(e) for e in g["entropies"]])
    torch.cuda.synchronize()
    acc = (preds.cpu().long() == torch.from_numpy(g["y_q"]).squeeze(2)).float().mean(1)
    print(f"{os.path.basename(path)[:-4]:28s} max|dW| {np.abs(w.cpu().numpy() - g['weights']).max():.2e}  "
          f"max|dlogit| {np.abs(lq.cpu().numpy() - g['logits_q']).max():.2e}  "
          f"crit rel {np.abs(crit[0].cpu().numpy() / g['criterions'] - 1).max():.2e}  "
          f"pred mismatches {(preds.cpu().numpy() != g['logits_q'].argmax(2)).sum()}  acc equal {np.array_equal(acc.numpy(), g['acc'][:, 0])}")

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fs_lshot_*.npz"))):
    g = np.load(path)
    unary, nbr, preds_iter, e = engine.run_laplacian_shot(
        torch.from_numpy(g["x_q"]).cuda(), torch.from_numpy(g["x_s"]).cuda(), torch.from_numpy(g["y_s"]).squeeze(2).cuda(),
        iters=int(g["iters"]), knn=int(g["knn"]), lmd=float(g["lmd"]), norm_type=str(g["norm_type"]))
    torch.cuda.synchronize()
    ref_u = g["unary"]
    du = np.abs(unary.cpu().numpy() - ref_u) / np.maximum(np.abs(ref_u), 1e-30)
    ee = e.cpu().numpy().reshape(g["ent_energy"].shape)
    de = np.abs(ee / g["ent_energy"] - 1)
    print(f"{os.path.basename(path)[:-4]:28s} unary rel {du.max():.2e}  energy rel {np.nanmax(de):.2e}  "
          f"neighbours equal {np.array_equal(np.sort(nbr.cpu().numpy(), axis=2), g['neighbours'])}  "
          f"preds equal {np.array_equal(preds_iter[:, -1].cpu().numpy(), g['preds'])}")

for K, N, shots, iters in ((100, 100, 4, 1000), (10, 100, 4, 1000), (397, 20, 4, 1000), (1000, 4, 4, 100)):
    x_q, _ = synth.make_query_tasks(N, K, seed=5, k_eff=5)
    x_s, y_s = synth.make_support(N, K, shots, seed=5)
    x_q, x_s, y_s = x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        engine.run_alpha_tim(x_q, x_s, y_s, iters=iters, temp=15.0, lr=1e-4, alpha_value=7.0)
        torch.cuda.synchronize()
        dt = time.time() - t0
    flop = 4.0 * N * (K * shots + 75) * K * K * iters
    print(f"K={K} N={N} shots={shots} iters={iters}: {dt:.3f} s  ({N / dt:.1f} tasks/s, {flop / dt / 1e12:.2f} TFLOP/s fp32 in the two GEMMs)")

for K, N, shots in ((100, 100, 4), (10, 100, 4), (397, 50, 4), (1000, 20, 4)):
    x_q, _ = synth.make_query_tasks(N, K, seed=5, k_eff=5)
    x_s, y_s = synth.make_support(N, K, shots, seed=5)
    x_q, x_s, y_s = x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        engine.run_laplacian_shot(x_q, x_s, y_s, iters=20, knn=3, lmd=0.7)
        torch.cuda.synchronize()
        dt = time.time() - t0
    print(f"LAPLACIAN_SHOT K={K} N={N} shots={shots}: {dt * 1e3:.1f} ms  ({N / dt:.0f} tasks/s)")
