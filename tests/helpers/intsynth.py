"""Platform-independent synthetic tasks for digest fixtures: probability features built from integer
draws and ONE correctly rounded float64 division per entry (no exp / softmax, whose bits depend on the
host's math library), so the fixture host and the GPU box generate the same float32 tensors."""
import numpy as np


def simplex_rows(rng, labels, n_class, boost=4096):
    n = labels.shape[0]
    raw = rng.integers(1, 1 << 16, size=(n, n_class)).astype(np.float64)
    raw[np.arange(n), labels] *= boost * (1 + rng.integers(0, 8, size=n))
    # a few near-ties and tiny entries, as CLIP features have
    raw[:, :: max(1, n_class // 3)] *= 1.0 / 64
    return (raw / raw.sum(1, keepdims=True)).astype(np.float32)


def make_tasks(seed, n_task, n_class, n_query=75, shots=0, boost=4096):
    """x_q (N,Q,K) f32, y_q (N,Q) i64 [, x_s (N,K*shots,K) f32, y_s (N,K*shots) i64].  `boost`: how far the labelled
    class stands out (4096: near one-hot rows; 64: soft rows on which the MM loop of a large batch still converges early)"""
    rng = np.random.default_rng(seed)
    x_q = np.empty((n_task, n_query, n_class), np.float32)
    y_q = np.empty((n_task, n_query), np.int64)
    for t in range(n_task):
        k_eff = int(rng.integers(min(3, n_class), min(10, n_class) + 1))
        classes = rng.permutation(n_class)[:k_eff]
        y = classes[rng.integers(0, k_eff, size=n_query)]
        x_q[t], y_q[t] = simplex_rows(rng, y, n_class, boost), y
    if not shots:
        return x_q, y_q
    y_s = np.repeat(np.arange(n_class), shots)
    x_s = np.stack([simplex_rows(rng, y_s, n_class, boost) for _ in range(n_task)])
    return x_q, y_q, x_s, np.tile(y_s, (n_task, 1))
