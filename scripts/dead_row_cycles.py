#!/usr/bin/env python3
"""Design study (CPU oracle): how long is the transient of a DEAD row (y = -10) before it is on its limit cycle, and how long is the
cycle?  Rows = the alpha rows one task holds after the first outer iteration (what dies after the first E-step starts there).
python scripts/dead_row_cycles.py K [n_rows] [iter_mm_first]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import numpy as np
from oracle import c_oracle
from tclip_amd import synth
K = int(sys.argv[1])
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else K
iter_mm = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
x, _ = synth.make_query_tasks(1, K, seed=3)
r = c_oracle.run(x.numpy(), iters=1, iter_mm=iter_mm, lambd=int(K / 5) * 75)
print("mm_iters", r["mm_iters"].tolist(), flush=True)
alpha = r["alpha"][0]
mus, ps = [], []
mu, p = ctypes.c_int32(), ctypes.c_int32()
for k in np.linspace(0, K - 1, n_rows).astype(int):
    row = np.ascontiguousarray(alpha[k])
    c_oracle.lib().tclip_oracle_dead_row_cycle(row.ctypes.data_as(ctypes.c_void_p), K, 400, ctypes.byref(mu), ctypes.byref(p))
    mus.append(mu.value); ps.append(p.value)
mus, ps = np.array(mus), np.array(ps)
print("rows", len(mus), "no cycle within 400:", int((mus < 0).sum()))
ok = mus >= 0
print("transient mu: min/median/mean/p90/p99/max", mus[ok].min(), np.median(mus[ok]), mus[ok].mean().round(1), np.percentile(mus[ok], 90), np.percentile(mus[ok], 99), mus[ok].max())
print("period: histogram", np.bincount(ps[ok]).tolist())
print("mu histogram (bins of 4):", np.bincount(mus[ok] // 4).tolist())
for L in (8, 12, 16, 20, 24, 32, 51):
    print(f"rows with mu <= {L}: {(mus[ok] <= L).mean():.3f}")
