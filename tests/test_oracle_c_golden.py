"""CPU: the C++ oracle (oracle/tclip_oracle.cpp) against the golden vectors the reference
produced.  Every special function and reduction order of torch CPU is restated bit for bit
(torch.log = MKL vsLn included), and on every fixture the oracle reproduces the reference's alpha, u and v EXACTLY,
with identical MM iteration counts and argmax traces."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from oracle import c_oracle

FAST = [n for n in golden_names() if any(f"K{k}_" in n for k in (2, 5, 6, 7, 10, 37)) and not n.startswith("eval_")]


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
    assert np.array_equal(out["mm_iters"], g["mm_iters"])
    assert np.array_equal(out["argmax"], g["argmax"])
    assert np.array_equal(out["alpha"], g["alpha"])
    assert np.array_equal(out["u"], g["u"])
    assert np.array_equal(out["v"], g["v"])
    assert np.array_equal(out["criterions"], g["criterions"])
