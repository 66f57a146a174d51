#!/usr/bin/env python3
"""SHA-1 digests of the outputs of the torch-eager restatement of the reference (oracle/ref_torch.py, which
tests/test_oracle_golden.py pins bit for bit to the reference itself) on seeded problems whose inputs are
generated platform-independently (tests/helpers/intsynth.py).  Run on the fixture host (torch 2.10 CPU,
AVX-512, MKL, 8 threads); tests/test_gpu_digests.py recomputes the same problems with the HIP engine on
the GPU box and compares digests - the special functions are thereby checked against TORCH, not against
the C++ oracle that shares csrc/tclip_math.h with the product.

    python tests/golden/make_digests.py        # writes tests/golden/digests_em_dirichlet.json
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import intsynth          # noqa: E402
from oracle import ref_torch          # noqa: E402


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


# (seed, K, N, Q, shots, hard, iters, iter_mm)
CASES = [
    (1, 2, 4, 75, 0, False, 3, 120), (2, 3, 3, 20, 0, True, 3, 101), (3, 5, 4, 75, 2, False, 3, 151),
    (4, 7, 3, 17, 0, False, 4, 60), (5, 8, 4, 75, 0, True, 3, 230), (6, 9, 2, 130, 1, True, 2, 120),
    (7, 17, 3, 75, 0, False, 3, 151), (8, 31, 2, 64, 0, False, 3, 101), (9, 32, 3, 75, 3, False, 3, 101),
    (10, 33, 2, 75, 0, True, 3, 120), (11, 40, 3, 5, 0, False, 4, 51), (12, 64, 2, 75, 0, False, 3, 151),
    (13, 65, 2, 75, 2, True, 3, 101), (14, 96, 2, 75, 0, False, 3, 120), (15, 100, 3, 75, 0, False, 4, 230),
    (16, 101, 2, 75, 4, False, 3, 101), (17, 129, 2, 1, 0, False, 2, 60), (18, 160, 1, 75, 0, True, 3, 151),
    (19, 200, 1, 75, 1, False, 3, 101), (20, 257, 1, 75, 0, False, 3, 120), (21, 300, 1, 75, 0, False, 2, 151),
    (22, 10, 4, 75, 0, False, 20, 1000), (23, 12, 3, 75, 2, False, 20, 1000), (24, 37, 2, 75, 0, True, 10, 1000),
]


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out = {"torch": torch.__version__, "threads": torch.get_num_threads(), "cases": []}
    for seed, K, N, Q, shots, hard, iters, iter_mm in CASES:
        t = intsynth.make_tasks(seed, N, K, Q, shots)
        x_q = torch.from_numpy(t[0])
        x_s = torch.from_numpy(t[2]) if shots else None
        y_s = torch.from_numpy(t[3]) if shots else None
        lambd = max(1, int(K / 5)) * Q
        r = ref_torch.run(x_q, x_s, y_s, n_class=K, iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
        out["cases"].append({"seed": seed, "K": K, "N": N, "Q": Q, "shots": shots, "hard": hard, "iters": iters,
                             "iter_mm": iter_mm, "lambd": lambd, "inputs": sha(t[0]) + (sha(t[2]) if shots else ""),
                             "alpha": sha(r["alpha"].numpy()), "u": sha(r["u"].numpy()), "v": sha(r["v"].numpy()),
                             "mm_iters": [int(m) for m in r["mm_iters"]]})
        print(out["cases"][-1], flush=True)
    with open(os.path.join(HERE, "digests_em_dirichlet.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
