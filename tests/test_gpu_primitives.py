"""GPU: the exact-rounding primitives, the branch-free MM update and the restated library
routines of csrc/tclip_math.h, checked ON THE DEVICE: mismatch counters against the compiler's
IEEE operators / generic routines must be zero, and the checksums of the routines over fixed
argument streams must equal those of the host build (which tests/test_math_host.py compares
with torch value by value)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def test_device_arithmetic_matches_ieee_and_host():
    import torch
    from oracle import build as oracle_build
    from tclip_amd import _capi
    torch.cuda.init()
    out = (ctypes.c_uint64 * 14)()
    _capi.check(_capi.lib().tclip_selftest_primitives(out), "tclip_selftest_primitives")
    vals = list(out)
    print("selftest:", vals, _capi.lib().tclip_last_error())
    rcp, div, digamma, lgamma, update = vals[:5]
    assert rcp == 0, f"{rcp} reciprocal mismatches over 3 exhaustive binades"
    assert div == 0, f"{div} quotient mismatches over 2^29 pairs"
    assert digamma == 0 and lgamma == 0, "fused digamma/lgamma differ from the generic routines"
    assert update == 0, f"{update} MM updates differ between the branch-free and the generic form"
    _, path = oracle_build.build()
    host = (ctypes.c_uint64 * 8)()
    ctypes.CDLL(path).mc_checksums(host)
    names = ["digamma", "lgamma", "sqrt", "exp", "log", "fused psi", "fused lgamma", "digamma_pos"]
    for n, h, d in zip(names, list(host), vals[6:14]):
        assert h == d, f"{n}: device checksum {d:#x} != host checksum {h:#x}"
