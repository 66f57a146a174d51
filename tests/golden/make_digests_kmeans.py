#!/usr/bin/env python3
"""SHA-1 digests of the torch-eager restatements (oracle/ref_torch.py, pinned to the reference's own classes by the
fixture tests of each method) of SOFT_KMEANS, HARD_KMEANS, EM_GAUSSIAN, EM_GAUSSIAN_COV, KL_KMEANS, PADDLE and BDCSPN on
seeded problems whose inputs are generated platform-independently (tests/helpers/intsynth.py).  Run on the fixture host
(torch 2.10 CPU, AVX-512, MKL, 8 threads); tests/test_gpu_digests_kmeans.py recomputes the same problems with the HIP
engine on the GPU box and compares digests - the logarithms, square roots and sums of these methods are thereby
checked against TORCH, not against an oracle that shares csrc/tclip_math.h with the product.  (Replaces the offline
pair scripts/gpu_dump_kmeans.py + scripts/check_kmeans_dump.py of rounds 1-2.)

    python tests/golden/make_digests_kmeans.py        # writes tests/golden/digests_kmeans.json
"""
import hashlib
import json
import os
import random
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
    a = a.detach().numpy() if torch.is_tensor(a) else a
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def cases(n_cases=32, seed=7):
    rng = random.Random(seed)
    out = []
    for case in range(n_cases):
        K = rng.choice([2, 3, 5, 7, 8, 9, 16, 21, 31, 32, 33, 47, 64, 65, 100, 101, 130, 200, 260])
        out.append({"case": case, "K": K, "N": rng.randint(1, 4), "iters": rng.randint(1, 8), "shots": rng.randint(1, 3),
                    "paddle_lambd": rng.choice([0.0, 3.0, 12.5]), "norm_type": ("UN", "L2N", "CL2N")[case % 3],
                    "boost": rng.choice([16, 256, 4096]), "data_seed": 100000 * seed + case})
    return out


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out = {"torch": torch.__version__, "threads": torch.get_num_threads(), "cases": []}
    for c in cases():
        x_q, y_q, x_s, y_s = intsynth.make_tasks(c["data_seed"], c["N"], c["K"], 75, c["shots"], boost=c["boost"])
        xq, xs, ys = torch.from_numpy(x_q), torch.from_numpy(x_s), torch.from_numpy(y_s)
        K, lam = c["K"], int(c["K"] / 5) * 75
        t = {"skm": ref_torch.run_soft_kmeans(xq, n_class=K, iters=c["iters"], temperature=30),
             "hkm": ref_torch.run_hard_kmeans(xq, n_class=K, iters=c["iters"]),
             "emg": ref_torch.run_em_gaussian(xq, n_class=K, iters=c["iters"], temperature=30, lambd=lam),
             "cov": ref_torch.run_em_gaussian_cov(xq, n_class=K, iters=c["iters"], lambd=lam),
             "klk": ref_torch.run_kl_kmeans(xq, n_class=K, iters=c["iters"]),
             "paddle": ref_torch.run_paddle(xq, xs, ys, n_class=K, iters=c["iters"], lambd=c["paddle_lambd"]),
             "bdcspn": ref_torch.run_bdcspn(xq, xs, ys, n_class=K, temp=30.0, norm_type=c["norm_type"])}
        want = {"skm": ("u", "w"), "hkm": ("u", "w"), "emg": ("u", "v", "w"), "cov": ("u", "v", "w", "s"), "klk": ("u", "w"),
                "paddle": ("u", "v", "w"), "bdcspn": ("prototypes", "u")}
        c["inputs"] = sha(x_q) + sha(x_s)
        c["digests"] = {m: {a: sha(t[m][a]) for a in arrs} for m, arrs in want.items()}
        out["cases"].append(c)
        print(c["case"], c["K"], c["N"], c["iters"], c["shots"], flush=True)
    with open(os.path.join(HERE, "digests_kmeans.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
