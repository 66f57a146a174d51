#!/usr/bin/env python3
"""Wavefront clocks per part of k_mm_split's iteration (libraries built with -DTCLIP_PHASE_CLOCK, scripts/build_variant.sh):

    TCLIP_LIB=gpurun_variants/clk_lazy.so python scripts/gpu_phase_clock.py [K B N iters hard shots] ...
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from tclip_amd import engine, synth, _capi

nums = [int(v) for v in sys.argv[1:]]
shapes = [tuple((nums[j:j + 6] + [0, 0])[:6]) for j in range(0, len(nums), 6)] or [(1000, 3, 125, 20, 0, 0), (100, 10, 100, 20, 0, 0)]
lib = _capi.lib()
fn = lib.tclip_debug_phase_clock
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64)]
names = ["head", "scatter/sort", "passes A", "passes B", "passes C", "phase C", "iterations", "abandoned"]
for K, B, N, iters, hard, shots in shapes:
    x_q, _ = synth.make_query_tasks(B * N, K, seed=3, k_eff=(5 if shots else None))
    x_q = x_q.cuda()
    x_s = y_s = None
    if shots:
        x_s, y_s = synth.make_support(B * N, K, shots, seed=3)
        x_s, y_s = x_s.cuda(), y_s.squeeze(2).cuda()
    out = (ctypes.c_uint64 * 8)()
    for rep in range(2):
        fn(out)
        torch.cuda.synchronize(); t = time.time()
        engine.run_em_dirichlet(x_q, x_s, y_s, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=bool(hard))
        torch.cuda.synchronize(); dt = time.time() - t
        fn(out)
    it = max(out[6], 1)
    tot = sum(out[i] for i in range(6))
    print(f"{os.environ.get('TCLIP_LIB', 'orig')}  K={K} B={B} N={N} iters={iters} hard={hard} shots={shots}: {dt:.3f}s  wave-iterations {out[6]}  abandoned {out[7]}  clocks/iteration {tot / it:.0f}")
    print("   " + "  ".join(f"{names[i]} {out[i] / it:.0f} ({100.0 * out[i] / tot:.1f} %)" for i in range(6)), flush=True)
