#!/usr/bin/env python3
"""Design study (CPU oracle): when do LIVE rows of the MM loop fall onto a fixed point or a short limit cycle of the fp32 map?
python scripts/live_row_cycles.py K N iters [max_period]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import numpy as np
from oracle import c_oracle
from tclip_amd import synth
K, N, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P = int(sys.argv[4]) if len(sys.argv) > 4 else 16
x, _ = synth.make_query_tasks(N, K, seed=3)
out = np.full((iters, N * K, 2), -2, np.int32)
c_oracle.lib().tclip_oracle_set_cycle_probe(out.ctypes.data_as(ctypes.c_void_p), P)
r = c_oracle.run(x.numpy(), iters=iters, iter_mm=1000, lambd=int(K / 5) * 75)
c_oracle.lib().tclip_oracle_set_cycle_probe(None, 0)
print("mm_iters", r["mm_iters"].tolist())
for it in range(iters):
    o = out[it]
    watched = o[:, 0] != -2
    # rows never touched keep -2 only if P == 0; dead rows keep -1/0 from the reset but were not watched: use period to tell
    hit = o[:, 0] >= 0
    n_mm = int(r["mm_iters"][it])
    saved = np.clip(n_mm - o[hit, 0], 0, None).sum()
    # live rows: those that were watched = hit or (not hit and live) - approximate the live count from u
    print(f"it {it}: rows on a cycle {int(hit.sum())}, median first-hit {np.median(o[hit,0]) if hit.any() else -1}, "
          f"periods {np.bincount(o[hit,1], minlength=4)[:8].tolist()}, row-iterations saved {int(saved)} of ~{n_mm} x live rows")
