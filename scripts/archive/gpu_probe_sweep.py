"""K=1000 wall time against the number of probed chunks: python3 scripts/gpu_probe_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth
x, _ = synth.make_query_tasks(250, 1000, seed=5); x = x.cuda()
for chunks in (0, 1, 2, 4, 19):
    engine.debug_set_probe_chunks(chunks)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        res = engine.run_em_dirichlet(x, n_batches=2, iters=20, iter_mm=1000, lambd=200 * 75, hard=False)
        torch.cuda.synchronize(); dt = time.time() - t
    print(f"probe_chunks={chunks}: {dt:.3f} s alpha_sum={res.alpha.double().sum().item():.9e}", flush=True)
