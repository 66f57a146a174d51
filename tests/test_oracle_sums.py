"""CPU: the reduction orders restated in oracle/tclip_oracle.cpp (and mirrored on the GPU in
csrc/tclip_device.h) against torch's own CPU kernels, bit for bit.  These orders are part of the
reference's arithmetic: torch.sum over the last / a strided dimension, softmax's denominator."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import c_oracle


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _avx_like():
    # the orders below are those of torch's AVX2/AVX-512 kernel set (8-float vectors for sum)
    cap = torch.backends.cpu.get_cpu_capability()
    return cap in ("AVX2", "AVX512")


pytestmark = pytest.mark.skipif(not _avx_like(), reason="reduction orders pinned for the AVX2/AVX-512 ATen kernels")


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 16, 17, 31, 32, 33, 37, 63, 64, 65, 75, 100, 127, 128, 129, 255,
                               256, 257, 397, 511, 512, 513, 1000, 1024, 4000])
def test_inner_and_outer_sum_order(n):
    lib = c_oracle.lib()
    torch.manual_seed(n)
    T = 64
    X = torch.randn(T, n) * torch.exp(torch.randn(T, n) * 3)
    inner = X.sum(-1).numpy()
    outer = X.t().contiguous().sum(0).numpy()
    for t in range(T):
        x = np.ascontiguousarray(X[t].numpy())
        assert lib.tclip_oracle_sum_inner(_p(x), ctypes.c_long(n)) == inner[t]
        assert lib.tclip_oracle_sum_outer(_p(x), ctypes.c_long(n), ctypes.c_long(t), ctypes.c_long(T)) == outer[t]


@pytest.mark.parametrize("shape", [(3, 75, 10), (2, 75, 37), (2, 75, 100), (2, 148, 37), (4, 75, 2), (4, 75, 3), (4, 75, 4),
                                   (4, 75, 5), (4, 75, 6), (4, 75, 7), (4, 75, 8), (4, 75, 9), (3, 20, 5), (3, 75, 21)])
def test_mstep_statistics_order(shape):
    """(u.unsqueeze(-1) * logz.unsqueeze(2)).sum(1) and u.sum(1), the reference's M-step sums."""
    lib = c_oracle.lib()
    N, Q, K = shape
    torch.manual_seed(K)
    A = torch.rand(N, Q, K, 1) * torch.exp(torch.randn(N, Q, K, 1) * 2)
    B = torch.randn(N, Q, 1, K)
    P = A * B
    S = P.sum(1)
    U = A.squeeze(-1)
    cs = U.sum(1)
    for n in range(N):
        for c in range(0, K * K, max(1, K * K // 53)):
            k, d = divmod(c, K)
            x = np.ascontiguousarray(P[n, :, k, d].numpy())
            assert lib.tclip_oracle_sum_outer(_p(x), ctypes.c_long(Q), ctypes.c_long(c), ctypes.c_long(K * K)) == S[n, k, d].item()
        for k in range(K):
            x = np.ascontiguousarray(U[n, :, k].numpy())
            assert lib.tclip_oracle_sum_outer(_p(x), ctypes.c_long(Q), ctypes.c_long(k), ctypes.c_long(K)) == cs[n, k].item()


@pytest.mark.parametrize("n", [3, 10, 15, 16, 17, 37, 100, 397, 1000])
def test_softmax_row(n):
    lib = c_oracle.lib()
    torch.manual_seed(n)
    X = torch.randn(50, n) * 4 + torch.randn(50, 1) * 100
    S = torch.softmax(X, 1).numpy()
    for t in range(50):
        x = np.ascontiguousarray(X[t].numpy())
        o = np.empty_like(x)
        lib.tclip_oracle_softmax_row(_p(x), _p(o), ctypes.c_long(n))
        assert np.array_equal(o, S[t])
