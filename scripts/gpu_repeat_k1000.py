"""Repeatability check at K=1000: python3 scripts/gpu_repeat_k1000.py [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from tclip_amd import engine, synth
x, _ = synth.make_query_tasks(250, 1000, seed=5); x = x.cuda()
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize(); t = time.time()
    res = engine.run_em_dirichlet(x, n_batches=2, iters=20, iter_mm=1000, lambd=200 * 75, hard=False)
    torch.cuda.synchronize(); dt = time.time() - t
    sig = (res.alpha.double().sum().item(), res.u.double().sum().item(), res.mm_iters.cpu().tolist())
    same = ref is None or (sig == ref)
    ref = ref or sig
    print(f"rep {rep}: {dt:.3f} s  alpha_sum={sig[0]:.9e} same_as_first={same} mm_iters={sig[2][0][:8]} {sig[2][1][:8]}", flush=True)
