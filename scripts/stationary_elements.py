#!/usr/bin/env python3
"""Design study (CPU oracle): what fraction of the elements of live rows does an MM iteration leave bitwise unchanged?
(An unchanged element's digamma(a+1), lgamma(a+1) and curvature are the previous iteration's.)
python scripts/stationary_elements.py K N iters"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import numpy as np
from oracle import c_oracle
from tclip_amd import synth
K, N, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x, _ = synth.make_query_tasks(N, K, seed=3)
out = np.zeros((iters, 1000, 2), np.int64)
c_oracle.lib().tclip_oracle_set_stationary_probe(out.ctypes.data_as(ctypes.c_void_p))
r = c_oracle.run(x.numpy(), iters=iters, iter_mm=1000, lambd=int(K / 5) * 75)
c_oracle.lib().tclip_oracle_set_stationary_probe(None)
print("mm_iters", r["mm_iters"].tolist())
for it in range(iters):
    n = int(r["mm_iters"][it])
    o = out[it, :n]
    frac = o[:, 0] / np.maximum(o[:, 1], 1)
    marks = [0, 1, 5, 10, 25, 50, 100, 200, 400, 600, 800, n - 1]
    print(f"it {it}: live elements {int(o[0, 1])}, unchanged fraction overall {o[:, 0].sum() / max(o[:, 1].sum(), 1):.3f}; at l = "
          + ", ".join(f"{l}: {frac[l]:.2f}" for l in marks if l < n))
