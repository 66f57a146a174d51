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


def test_lgamma_bit_exact_vs_torch(mc):
    """torch.lgamma = Sleef lgammaf_u10, restated in float-float arithmetic: every 5th float of
    [2^-20, 16) (reflection, both polynomial branches, the shifted Stirling branch) and random
    arguments beyond, bit for bit.  (The full exhaustive sweep of the range was run once when the
    restatement was written: 0 mismatches over 2.0e8 floats.)"""
    lo, hi = np.float32(2.0 ** -20).view(np.int32), np.float32(16.0).view(np.int32)
    x = np.arange(lo, hi, 5, dtype=np.int32).view(np.float32)
    assert np.array_equal(_call(mc, "mc_lgamma", x), torch.lgamma(torch.from_numpy(x.copy())).numpy())
    for lo_, hi_ in [(1e-37, 2.0 ** -20), (16, 1e30)]:
        y = _loguniform(lo_, hi_, 1 << 21, 7)
        assert np.array_equal(_call(mc, "mc_lgamma", y.numpy()), torch.lgamma(y).numpy())
    a = _loguniform(1e-12, 1e7, 1 << 21, 8)
    assert np.array_equal(_call(mc, "mc_xp1_lg", a.numpy()), torch.lgamma(a + 1).numpy())


def test_sqrt_bit_exact_vs_torch(mc):
    """torch.sqrt on this (AVX-512) host = MKL vsSqrt HA = VRSQRT14PS + one Heron step, which is
    not the IEEE square root; restated with a table of the instruction's values."""
    if torch.backends.cpu.get_cpu_capability() != "AVX512":
        pytest.skip("MKL dispatches another vsSqrt kernel on hosts without AVX-512")
    x = _loguniform(1e-12, 1e12, 1 << 22, 9)
    ours, ref = _call(mc, "mc_sqrt_torch", x.numpy()), torch.sqrt(x).numpy()
    assert np.array_equal(ours, ref)
    assert 0.003 < (ref != np.sqrt(x.numpy())).mean() < 0.012     # ... and torch's is indeed not IEEE


@pytest.mark.skipif(torch.backends.cpu.get_cpu_capability() != "AVX512",
                    reason="the restated kernel is MKL's AVX-512 vsLn; other hosts dispatch another one")
def test_log_bit_exact_vs_torch(mc):
    """MKL vsLn (HA, AVX-512 kernel) restated from its disassembly: every float of [0.5, 2) - where
    MKL is one ulp off the correctly rounded value on 1e-3 of arguments - and 4e6 log-uniform
    arguments over the rest of the restated range."""
    x = np.arange(0x3f000000, 0x40000000, dtype=np.uint32).view(np.float32)
    assert np.array_equal(_call(mc, "mc_log", x).view(np.uint32), torch.log(torch.from_numpy(x)).numpy().view(np.uint32))
    for lo, hi, seed in ((1e-30, 1e-15, 5), (1e-15, 0.5, 6), (2.0, 1e30, 7)):
        y = _loguniform(lo, hi, 1 << 21, seed)
        assert np.array_equal(_call(mc, "mc_log", y.numpy()).view(np.uint32), torch.log(y).numpy().view(np.uint32))


def test_exp_matches_torch_softmax_exp(mc):
    """Sleef expf_u10 as torch's CPU softmax uses it: softmax([0, x]) with x <= -17.5 returns
    exp(x) exactly (the sum is 1.0f), which isolates the exponential."""
    g = torch.Generator().manual_seed(6)
    x = -(torch.rand(1 << 18, generator=g) * (104 - 17.5) + 17.5)
    rows = torch.stack([torch.zeros_like(x), x], 1)
    u = torch.softmax(rows, 1)
    assert (u[:, 0] == 1).all()
    assert np.array_equal(_call(mc, "mc_exp", x.numpy()), u[:, 1].numpy())


def test_lgamma_fp64_form_agrees_on_every_float_of_1_to_2p3(mc):
    """the cheaper evaluation of lgamma on [1, 2.3) used inside the MM kernel (fp64 tail with a
    'sure' filter) against the double-float restatement of Sleef, exhaustively"""
    out = (ctypes.c_ulonglong * 3)()
    mc.mc_lgamma_f64_form(out)
    differ, unsure, visited = list(out)
    assert visited == 9646899          # every float in [1, 2.3)
    assert differ == 0
    assert 0 < unsure < visited // 10000


def test_lgamma_fp64_form_agrees_on_every_float_of_2p3_to_2e41(mc):
    """the fp64 evaluation of Sleef's large-argument lgamma used by the MM kernel's dense pass against
    the double-float restatement (itself bit-exact against torch above), on EVERY float of
    [2.3, 2^41] - the whole domain the kernel can feed it (alpha + 1 with alpha <= 2^40)"""
    import struct
    bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
    out = (ctypes.c_ulonglong * 4)()
    mc.mc_lgamma_ge23_f64_form(ctypes.c_uint(bits(2.3)), ctypes.c_uint(bits(2.0 ** 41) + 1), ctypes.c_uint(1), out)
    differ, unsure, visited, needed_window = list(out)
    assert visited == bits(2.0 ** 41) + 1 - bits(2.3)          # 334 286 030 floats
    assert differ == 0, "a 'sure' argument rounds differently in the fp64 form"
    assert 0 < unsure < visited // 10000
    assert needed_window <= 1 << 10, needed_window             # the window in use is 2^13


def test_no_shift_lgamma_form_agrees_on_every_float_above_7(mc):
    """class C of the split MM kernel (alpha + 1 >= 10) evaluates Sleef's large-argument lgamma without the argument shift
    and with the logarithm of the Stirling correction as a six-term log1p series (lgamma_sleef_gt7_f64): against the double-float restatement
    on EVERY float of (7, 2^41]"""
    import struct
    bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
    out = (ctypes.c_ulonglong * 4)()
    mc.mc_lgamma_gt7_f64_form(ctypes.c_uint(bits(7.0) + 1), ctypes.c_uint(bits(2.0 ** 41) + 1), ctypes.c_uint(1), out)
    differ, unsure, visited, needed_window = list(out)
    assert visited == bits(2.0 ** 41) - bits(7.0)
    assert differ == 0, "a 'sure' argument rounds differently in the no-shift form"
    assert 0 < unsure < visited // 10000
    assert needed_window <= 1 << 10, needed_window             # measured below; the window in use is 2^13


def test_digamma_recurrence_pieces_on_every_float_of_1_to_24(mc):
    """the split MM kernel evaluates digamma(alpha + 1) in pieces: the recurrence's partial sum on a dense queue, where the
    recurrence leaves x in closed form (digamma_rec_x), the series afterwards.  The closed form against the loop on EVERY
    float of [1, 24); the recomposed digamma and the class-wise recurrences on every eighth."""
    out = (ctypes.c_ulonglong * 4)()
    mc.mc_rec_closed_form(out)
    bad_x, bad_psi, bad_lg, visited = list(out)
    assert visited == 37748736
    assert bad_x == 0 and bad_psi == 0 and bad_lg == 0


def test_sqrt_on_the_derived_table_equals_the_vrsqrt14_restatement(mc):
    """the MM kernels read a derived VRSQRT14 table (csrc/tclip_rsqrt14_table_dev.h: 32-bit entries, indexed by the argument's
    bits as they stand) and skip the instruction's exact result at the powers of 4; torch.sqrt's Heron step gives the same
    root from either estimate.  Every float of [1, 4) at five scales and every power of two of [2^-100, 2^100], and the
    committed header is what tools/gen_rsqrt14_dev_table.py derives from the instruction's table."""
    import subprocess
    import sys
    out = (ctypes.c_ulonglong * 3)()
    mc.mc_sqrt_without_pow4(out)
    differ, visited, powers_of_4 = list(out)
    assert visited == 5 * (1 << 24) + 201 and powers_of_4 == 1 + 101
    assert differ == 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.call([sys.executable, os.path.join(root, "tools", "gen_rsqrt14_dev_table.py"), "--check"]) == 0
