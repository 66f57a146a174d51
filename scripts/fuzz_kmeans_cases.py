"""Seeded problem list shared by scripts/gpu_dump_kmeans.py and scripts/check_kmeans_dump.py."""
import random


def cases(n_cases, seed):
    rng = random.Random(seed)
    out = []
    for case in range(n_cases):
        K = rng.choice([2, 3, 5, 7, 8, 9, 16, 21, 31, 32, 33, 47, 64, 65, 100, 101, 130, 200, 260])
        out.append({"case": case, "K": K, "N": rng.randint(1, 5), "iters": rng.randint(1, 8), "shots": rng.randint(1, 3),
                    "paddle_lambd": rng.choice([0.0, 3.0, 12.5]), "data_seed": 100000 * seed + case})
    return out
