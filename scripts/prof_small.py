"""Small fixed workload for counter profiling: python3 scripts/prof_small.py [K] [B] [N] [iters] [hard] [shots]
(shots > 0: few-shot, shots support rows per class and task)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
hard = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
shots = int(sys.argv[6]) if len(sys.argv) > 6 else 0
if os.environ.get('TCLIP_WIDE'):
    engine.debug_set_rowset_min_rows(0)          # 32 lanes per row for every K (test hook)
if os.environ.get('TCLIP_SPLIT_MODE'):
    from tclip_amd import _capi
    _capi.lib().tclip_debug_set_mm_split(int(os.environ['TCLIP_SPLIT_MODE']))      # 0: k_mm_live everywhere (the round-2 path)
x_q, y_q = synth.make_query_tasks(B * N, K, seed=3)
x_q = x_q.cuda()
x_s = y_s = None
if shots:
    x_s, y_s = synth.make_support(B * N, K, shots, seed=4)
    x_s, y_s = x_s.cuda(), y_s.squeeze(2).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    engine.profile_enable(True)
    res = engine.run_em_dirichlet(x_q, x_s, y_s, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=hard)
    ms, ms_sum, n, upd = engine.profile_collect()
    engine.profile_enable(False)
    print(f"K={K} B={B} N={N} iters={iters} total={time.time()-t:.3f}s mm_ms={ms:.1f} launches={n} updates={upd:.3e} "
          f"updates/s={upd/ms*1e3:.3e} mm_iters={res.mm_iters[0].tolist()}", flush=True)
