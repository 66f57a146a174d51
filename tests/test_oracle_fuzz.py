"""CPU: the C++ oracle against the torch-eager restatement of the reference on random small
problems (class counts on both sides of every vector-width threshold of ATen's reductions, odd
query counts, few-shot, hard).  The two share nothing but the algorithm: the torch side uses
torch's own kernels, the C++ side the restated special functions and reduction orders."""
import random

import numpy as np
import pytest
import torch

from oracle import c_oracle, ref_torch
from tclip_amd import synth

pytestmark = pytest.mark.skipif(torch.backends.cpu.get_cpu_capability() != "AVX512",
                                reason="reduction orders are pinned for torch's AVX-512 host kernels")


def _cases():
    rng = random.Random(2024)
    out = []
    for K in (2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 40):
        out.append((K, rng.choice([20, 33, 75]), rng.randint(2, 3), rng.random() < 0.35, rng.random() < 0.4,
                    rng.choice([51, 60, 101]), rng.randint(2, 3)))
    return out


@pytest.mark.parametrize("K,Q,N,few,hard,iter_mm,iters", _cases())
def test_c_oracle_equals_torch_restatement(K, Q, N, few, hard, iter_mm, iters):
    lambd = max(1, int(K / 5)) * Q
    x_q, _ = synth.make_query_tasks(N, K, seed=700 + K, n_query=Q, k_eff=(min(3, K) if few else None))
    x_s = y_s = None
    if few:
        x_s, y_s = synth.make_support(N, K, 2, seed=800 + K)
    c = c_oracle.run(x_q.numpy(), x_s.numpy() if few else None, y_s.numpy() if few else None,
                     iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
    t = ref_torch.run(x_q, x_s, y_s, n_class=K, iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
    assert np.array_equal(c["mm_iters"], np.asarray(t["mm_iters"]))
    assert np.array_equal(c["alpha"], t["alpha"].numpy())
    assert np.array_equal(c["u"], t["u"].numpy())
    assert np.array_equal(c["v"], t["v"].numpy())


@pytest.mark.parametrize("K", [2, 3, 4, 5, 7, 8, 9, 16, 21, 33])
def test_kmeans_family_c_oracle_equals_torch_restatement(K):
    x_q, _ = synth.make_query_tasks(3, K, seed=900 + K, k_eff=min(4, K))
    c = c_oracle.run_soft_kmeans(x_q.numpy(), iters=4, temperature=30)
    t = ref_torch.run_soft_kmeans(x_q, n_class=K, iters=4, temperature=30)
    assert np.array_equal(c["u"], t["u"].numpy()) and np.array_equal(c["w"], t["w"].numpy())
