"""GPU: the correctly rounded fp32 primitives and the fused digamma/lgamma routine of
csrc/tclip_math.h, checked on the device against the compiler's IEEE operators."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def test_fast_primitives_are_correctly_rounded():
    import torch
    from tclip_amd import _capi
    torch.cuda.init()
    out = (ctypes.c_uint64 * 7)()
    _capi.check(_capi.lib().tclip_selftest_primitives(out), "tclip_selftest_primitives")
    rcp, sqrt, div, digamma, lgamma, rcp1, update = list(out)
    print(f"selftest counters: rcp={rcp} sqrt={sqrt} div={div} digamma={digamma} lgamma={lgamma} rcp_one_step={rcp1} update={update}")
    assert rcp == 0, f"{rcp} reciprocal mismatches over 3 exhaustive binades"
    assert sqrt == 0, f"{sqrt} square-root mismatches over 2 x 2^24 arguments"
    assert div == 0, f"{div} quotient mismatches over 2^29 pairs"
    assert digamma == 0, f"{digamma} digamma mismatches between the fused and the generic routine"
    assert lgamma <= 2 ** 24 * 1e-4, f"{lgamma} lgamma differences (expected: rare fp64 double-rounding cases)"
    assert update <= 2 ** 24 * 1e-4, f"{update} whole-update differences between the branch-free and the generic form"
