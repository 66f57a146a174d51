"""CPU: host build of the product's special-function header (csrc/tclip_math.h) against torch's
CPU implementations, value by value."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import build as oracle_build


@pytest.fixture(scope="module")
def mc():
    _, path = oracle_build.build()
    return ctypes.CDLL(path)


def _call(lib, name, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    getattr(lib, name)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    return y


def _loguniform(lo, hi, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.exp(torch.rand(n, dtype=torch.float64, generator=g) * (np.log(hi) - np.log(lo)) + np.log(lo)).float()


RANGES = [(1e-6, 1), (1, 2.5), (2.5, 10), (10, 30), (30, 1e8)]


@pytest.mark.parametrize("lo,hi", RANGES)
def test_digamma_bit_exact_vs_torch(mc, lo, hi):
    """calc_digamma(float) of ATen (torch.polygamma(0, .) on CPU): every bit."""
    x = _loguniform(lo, hi, 1 << 20, 1)
    ref = torch.digamma(x).numpy()
    assert np.array_equal(_call(mc, "mc_digamma", x.numpy()), ref)
    assert np.array_equal(_call(mc, "mc_digamma_pos", x.numpy()), ref)


@pytest.mark.parametrize("lo,hi", [(1e-12, 1e-3), (1e-3, 1), (1, 9), (8.9, 9.1), (9, 1e7)])
def test_fused_digamma_of_alpha_plus_one_bit_exact(mc, lo, hi):
    a = _loguniform(lo, hi, 1 << 20, 2)
    assert np.array_equal(_call(mc, "mc_xp1_psi", a.numpy()), torch.digamma(a + 1).numpy())


def test_logf_matches_libm(mc):
    """glibc logf (what calc_digamma calls) restated in fp64: identical on this host."""
    x = _loguniform(1e-30, 1e30, 1 << 21, 3)
    assert np.array_equal(_call(mc, "mc_logf_glibc", x.numpy()), _call(mc, "mc_libm_logf", x.numpy()))


def test_lgamma_is_correctly_rounded_and_close_to_torch(mc):
    """lgamma: the correctly rounded value (vs glibc's fp64 lgamma rounded once); torch's Sleef
    lgammaf_u10 differs from it by 1 ulp on a measured share of arguments (the parity residue)."""
    for lo, hi, max_cr, max_sleef in [(1, 2.5, 1e-4, 0.25), (2.5, 10, 1e-4, 0.01), (10, 1e6, 1e-4, 2e-3)]:
        x = _loguniform(lo, hi, 1 << 20, 4)
        ours = _call(mc, "mc_lgamma", x.numpy())
        assert (ours != _call(mc, "mc_lgamma_cr", x.numpy())).mean() <= max_cr
        t = torch.lgamma(x).numpy()
        diff = ours != t
        assert diff.mean() <= max_sleef
        assert np.abs(ours.view(np.int32).astype(np.int64) - t.view(np.int32).astype(np.int64)).max() <= 1
        a = x - 1
        fused = _call(mc, "mc_xp1_lg", a.numpy())
        ok = a.numpy() >= 2.0 ** -10
        assert (fused[ok] != _call(mc, "mc_lgamma", (a + 1).numpy())[ok]).mean() <= 1e-4


def test_log_close_to_torch(mc):
    x = _loguniform(1e-12, 1.0, 1 << 20, 5)
    assert (_call(mc, "mc_log", x.numpy()) != torch.log(x).numpy()).mean() <= 2e-3


def test_exp_matches_torch_softmax_exp(mc):
    """Sleef expf_u10 as torch's CPU softmax uses it: softmax([0, x]) with x <= -17.5 returns
    exp(x) exactly (the sum is 1.0f), which isolates the exponential."""
    g = torch.Generator().manual_seed(6)
    x = -(torch.rand(1 << 18, generator=g) * (104 - 17.5) + 17.5)
    rows = torch.stack([torch.zeros_like(x), x], 1)
    u = torch.softmax(rows, 1)
    assert (u[:, 0] == 1).all()
    assert np.array_equal(_call(mc, "mc_exp", x.numpy()), u[:, 1].numpy())
