#!/usr/bin/env python3
"""Search for seeded problems whose MM stop test lands very close to its threshold (1e-11), using the
C++ oracle (bit-identical to the reference on alpha, milliseconds per small run).  A hit is then handed to
make_golden.py, which runs the REFERENCE on it and records the fp32 norms the reference saw.

    python tests/golden/find_borderline.py [n_seeds]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
from oracle import c_oracle          # noqa: E402
from tclip_amd import synth          # noqa: E402

margin = c_oracle.lib().tclip_oracle_min_stop_margin
margin.restype = ctypes.c_double
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
best = []
for seed in range(5000, 5000 + n):
    K, N = 5 + seed % 6, 2 + seed % 3
    x_q, _ = synth.make_query_tasks(N, K, seed=seed)
    margin(1)
    c_oracle.run(x_q.numpy(), iters=20, iter_mm=1000, lambd=int(K / 5) * 75)
    m = margin(1)
    best.append((m, seed, K, N))
    if m < 1e-4:
        print("hit", m, seed, K, N, flush=True)
best.sort()
print(best[:5])
