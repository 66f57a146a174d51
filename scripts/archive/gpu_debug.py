import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
from oracle import c_oracle
from tclip_amd import engine, synth
K, N, B, hard = [int(v) for v in sys.argv[1:5]]
iters, iter_mm = int(sys.argv[5]), int(sys.argv[6])
x_q, _ = synth.make_query_tasks(B * N, K, seed=100 + K)
res = engine.run_em_dirichlet(x_q.cuda(), n_batches=B, iters=iters, iter_mm=iter_mm, lambd=int(K / 5) * 75, hard=bool(hard))
torch.cuda.synchronize()
for b in range(B):
    sl = slice(b * N, (b + 1) * N)
    ref = c_oracle.run(x_q[sl].numpy(), iters=iters, iter_mm=iter_mm, lambd=int(K / 5) * 75, hard=bool(hard))
    a = res.alpha[sl].cpu().numpy(); d = a != ref["alpha"]
    print("batch", b, "mm", res.mm_iters[b].tolist(), ref["mm_iters"].tolist(), "alpha neq", d.sum(), "of", d.size,
          "nan", np.isnan(a).sum(), "preds eq", np.array_equal(res.preds[sl].cpu().numpy(), ref["argmax"][-1]))
    if d.any():
        idx = np.argwhere(d)
        rows = sorted(set((int(i[0]), int(i[1])) for i in idx))
        print("   rows differing:", rows[:12], "...", len(rows))
        n, k = rows[0]
        print("   gpu", a[n, k][:8], "\n   ref", ref["alpha"][n, k][:8])
