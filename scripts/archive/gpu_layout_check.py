"""default layout against the forced 32-lane layout on one K: python3 scripts/gpu_layout_check.py K [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth
K = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x_q, _ = synth.make_query_tasks(2 * N, K, seed=70 + K)
runs = []
for wide in (-1, 0):
    engine.debug_set_rowset_min_rows(wide)
    r = engine.run_em_dirichlet(x_q.cuda(), n_batches=2, iters=3, iter_mm=230, lambd=int(K / 5) * 75)
    torch.cuda.synchronize(); runs.append(r)
engine.debug_set_rowset_min_rows(-1)
print(K, "equal:", torch.equal(runs[0].alpha, runs[1].alpha), torch.equal(runs[0].u, runs[1].u), torch.equal(runs[0].mm_iters, runs[1].mm_iters))
