#!/usr/bin/env python3
"""Design study: how many wavefront-iterations / block-iterations of k_mm_live queue nothing for the large-argument lgamma (library built
with -DTCLIP_COUNT_SMALL):  TCLIP_LIB=gpurun_variants/cnt.so python scripts/gpu_small_count.py [K B N iters hard shots] ..."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from tclip_amd import engine, synth, _capi
nums = [int(v) for v in sys.argv[1:]]
shapes = [tuple((nums[j:j + 6] + [0, 0])[:6]) for j in range(0, len(nums), 6)] or [(1000, 2, 125, 1, 0, 0), (1000, 2, 25, 1, 0, 4), (397, 2, 100, 1, 1, 0)]
fn = _capi.lib().tclip_debug_small_count
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64)]
engine.debug_set_probe_chunks(0)          # no dead-row shortcuts: with iters = 1 there are no dead rows anyway
for K, B, N, iters, hard, shots in shapes:
    x_q, _ = synth.make_query_tasks(B * N, K, seed=3, k_eff=(5 if shots else None)); x_q = x_q.cuda()
    x_s = y_s = None
    if shots:
        x_s, y_s = synth.make_support(B * N, K, shots, seed=3); x_s, y_s = x_s.cuda(), y_s.squeeze(2).cuda()
    out = (ctypes.c_uint64 * 4)()
    fn(out)
    r = engine.run_em_dirichlet(x_q, x_s, y_s, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=bool(hard))
    torch.cuda.synchronize()
    fn(out)
    print(f"K={K} B={B} N={N} iters={iters} hard={hard} shots={shots} mm_iters {r.mm_iters[0].tolist()}: wavefront-iterations {out[0]}, queued nothing {out[1]} "
          f"({100.0 * out[1] / max(out[0], 1):.1f} %); block-iterations {out[2]}, no wavefront queued anything {out[3]} ({100.0 * out[3] / max(out[2], 1):.1f} %)", flush=True)
