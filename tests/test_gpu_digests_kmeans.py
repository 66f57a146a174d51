"""GPU: the k-means family (SOFT/HARD_KMEANS, EM_GAUSSIAN, EM_GAUSSIAN_COV, KL_KMEANS), PADDLE and BDCSPN against digests
of the torch-eager restatement of the reference made on the fixture host (tests/golden/make_digests_kmeans.py):
32 seeded problems (K = 2 .. 260, 1 .. 4 tasks, 1 .. 8 iterations, 1 .. 3 shots, UN / L2N / CL2N) whose inputs are built
from integer draws and one division per entry, so this box regenerates the same tensors.  Every output array must
have torch's bits: the restated MKL logarithm, ATen's norm and sum orders and MKL's sgemm chains are checked against
torch itself, not against an oracle that shares csrc/tclip_math.h with the product."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sha(t):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else t
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases():
    p = os.path.join(GOLDEN, "digests_kmeans.json")
    return json.load(open(p))["cases"] if os.path.exists(p) else []


def test_digest_fixture_present():
    assert len(_cases()) >= 30


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"case{c['case']}_K{c['K']}")
def test_kmeans_family_matches_torch_digests(c):
    from helpers import intsynth
    from tclip_amd import engine
    x_q, y_q, x_s, y_s = intsynth.make_tasks(c["data_seed"], c["N"], c["K"], 75, c["shots"], boost=c["boost"])
    assert _sha(x_q) + _sha(x_s) == c["inputs"], "input generator is not reproducible on this host"
    xq, xs, ys = torch.from_numpy(x_q).to(DEV), torch.from_numpy(x_s).to(DEV), torch.from_numpy(y_s).to(DEV)
    K, lam = c["K"], int(c["K"] / 5) * 75
    got = {}
    u, w, _ = engine.run_soft_kmeans(xq, iters=c["iters"], temperature=30)
    got["skm"] = {"u": u, "w": w}
    u, w, _, _ = engine.run_hard_kmeans(xq, iters=c["iters"])
    got["hkm"] = {"u": u, "w": w}
    u, v, w, _ = engine.run_em_gaussian(xq, iters=c["iters"], temperature=30, lambd=lam)
    got["emg"] = {"u": u, "v": v, "w": w}
    u, v, w, s, _ = engine.run_em_gaussian_cov(xq, iters=c["iters"], lambd=lam)
    got["cov"] = {"u": u, "v": v, "w": w, "s": s}
    u, w, _, _ = engine.run_kl_kmeans(xq, iters=c["iters"])
    got["klk"] = {"u": u, "w": w}
    u, v, w, _ = engine.run_paddle(xq, xs, ys, iters=c["iters"], lambd=c["paddle_lambd"])
    got["paddle"] = {"u": u, "v": v, "w": w}
    pr, u, _ = engine.run_bdcspn(xq, xs, ys, temp=30.0, norm_type=c["norm_type"])
    got["bdcspn"] = {"prototypes": pr, "u": u}
    torch.cuda.synchronize()
    bad = [f"{m}.{a}" for m, arrs in c["digests"].items() for a, h in arrs.items() if _sha(got[m][a]) != h]
    assert not bad, f"K={c['K']} N={c['N']} iters={c['iters']} shots={c['shots']} {c['norm_type']}: differ from torch: {bad}"
