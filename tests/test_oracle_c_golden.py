"""CPU: the C++ oracle (oracle/tclip_oracle.cpp) against the golden vectors the reference
produced.  It is not bit-exact (torch's lgamma/log are replaced by correctly rounded values):
argmax and accuracies exact, MM counts exact up to borderline stop decisions, alpha within
ALPHA_TOL in the per-task Frobenius sense."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from oracle import c_oracle

ALPHA_TOL = 2e-5       # see tests/test_gpu_parity_golden.py for the measured figures
FAST = [n for n in golden_names() if ("K10_" in n or "K37_" in n) and not n.startswith("eval_")]


def _borderline(g):
    st = g["stop_test"].astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        crit = (st[:, :, 0] ** 2) / (st[:, :, 1] ** 2)
    return set(np.nonzero((np.abs(crit / np.float32(1e-11) - 1.0) < 0.02).any(axis=1))[0].tolist())


@pytest.mark.parametrize("name", FAST)
def test_c_oracle_vs_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kind = str(g["kind"])
    few = kind.startswith("fs")
    K = int(g["K"])
    out = c_oracle.run(g["x_q"], g["x_s"] if few else None, g["y_s"] if few else None, iters=int(g["iters"]),
                       iter_mm=int(g["iter_mm"]), lambd=int(K / 5) * 75, hard=kind.endswith("hard"))
    diff = set(np.nonzero(out["mm_iters"] != g["mm_iters"])[0].tolist())
    assert diff <= _borderline(g)
    assert np.array_equal(out["argmax"], g["argmax"])
    a, ref = out["alpha"].astype(np.float64), g["alpha"].astype(np.float64)
    fro = np.sqrt(((a - ref) ** 2).sum((1, 2))) / np.sqrt((ref ** 2).sum((1, 2)))
    assert fro.max() <= (4e-5 if diff else ALPHA_TOL)
    assert np.abs(out["u"] - g["u"]).max() <= 1e-5
    if not few:
        assert np.array_equal(out["v"], g["v"])     # log + cascade sums: bit-exact
