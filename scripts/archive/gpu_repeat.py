"""Run-to-run repeatability of the engine: python3 scripts/gpu_repeat.py K n_batches tasks_per_batch [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth
K, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
x, _ = synth.make_query_tasks(B * N, K, seed=5); x = x.cuda()
ref = None
for rep in range(reps):
    torch.cuda.synchronize(); t = time.time()
    res = engine.run_em_dirichlet(x, n_batches=B, iters=20, iter_mm=1000, lambd=int(K / 5) * 75, hard=False)
    torch.cuda.synchronize(); dt = time.time() - t
    if ref is None:
        ref = res
    same = torch.equal(res.alpha, ref.alpha) and torch.equal(res.u, ref.u) and torch.equal(res.mm_iters, ref.mm_iters)
    print(f"rep {rep}: {dt:.3f} s identical_to_first={same} mm_iters[0][:6]={res.mm_iters[0][:6].tolist()}", flush=True)
