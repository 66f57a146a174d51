"""GPU: the HIP engine (through the C ABI) against the golden vectors produced by the reference.

North star: argmax bit-exact; alpha / responsibilities within 1e-5 relative (fp32).
Measured on MI355X since the special functions are restated bit for bit (round 1): on all 14
fixtures (K = 10..1000, zero-/few-shot, soft/hard) alpha, u and v are IDENTICAL to the reference's,
MM iteration counts and accuracies included.  The assertions below therefore demand equality on
the small fixtures and keep 1e-6 of slack only where full tensors are not stored.  torch.log
(MKL vsLn) is restated too since its 1-ulp deviations from the correctly rounded log (2e-4 of
probability-like arguments) turned up in randomised sweeps.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names

pytestmark = pytest.mark.gpu
ALPHA_TOL = 1e-6      # large fixtures (sampled rows / checksums); measured: 0
SMALL = [n for n in golden_names() if "K397" not in n and "K1000" not in n]
LARGE = [n for n in golden_names() if "K397" in n or "K1000" in n]


def _run(g):
    from tclip_amd import engine
    kind = str(g["kind"])
    few = kind.startswith("fs")
    dev = torch.device("cuda:0")
    K = int(g["K"])
    res = engine.run_em_dirichlet(
        torch.from_numpy(g["x_q"]).to(dev),
        torch.from_numpy(g["x_s"]).to(dev) if few else None,
        torch.from_numpy(g["y_s"]).to(dev) if few else None,
        n_batches=1, iters=int(g["iters"]), iter_mm=int(g["iter_mm"]), lambd=int(K / 5) * 75,
        hard=kind.endswith("hard"))
    torch.cuda.synchronize()
    return res


def _borderline_iterations(g):
    """outer iterations whose stop decision in the reference was within 2 % of the threshold"""
    if "stop_test" not in g.files:
        return set()
    st = g["stop_test"].astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        crit = (st[:, :, 0] ** 2) / (st[:, :, 1] ** 2)
    near = np.abs(crit / np.float32(1e-11) - 1.0) < 0.02
    return set(np.nonzero(near.any(axis=1))[0].tolist())


def _fro_rel(a, ref):
    a = a.reshape(a.shape[0], -1).astype(np.float64)
    ref = ref.reshape(ref.shape[0], -1).astype(np.float64)
    return np.sqrt(((a - ref) ** 2).sum(1)) / np.sqrt((ref ** 2).sum(1))


@pytest.mark.parametrize("name", SMALL)
def test_engine_matches_reference_golden(name):
    from tclip_amd import engine
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    few = str(g["kind"]).startswith("fs")
    res = _run(g)
    assert np.array_equal(res.mm_iters.cpu().numpy()[0], g["mm_iters"]), "MM iteration counts differ"
    preds = res.preds.cpu().numpy()
    assert np.array_equal(preds, g["argmax"][-1].astype(np.int32)), "final argmax differs"
    fro = _fro_rel(res.alpha.cpu().numpy(), g["alpha"])
    assert np.array_equal(res.alpha.cpu().numpy(), g["alpha"]), f"alpha differs (Frobenius-relative {fro.max():.2e})"
    assert np.array_equal(res.u.cpu().numpy(), g["u"]), "responsibilities differ"
    assert np.array_equal(res.v.cpu().numpy(), g["v"]), "v differs"
    crit = res.criterions.cpu().numpy()[0]
    # the engine accumulates the norms in fp64, the reference in fp32
    np.testing.assert_allclose(crit, g["criterions"], rtol=2e-6, atol=1e-9)
    y_q = torch.from_numpy(g["y_q"]).squeeze(2)
    if few:
        acc = (res.preds.cpu().long() == y_q).float().mean(1, keepdim=True).numpy()
    else:
        acc_t, _ = engine.clustering_accuracy(torch.from_numpy(g["x_q"]).cuda(), res.preds, y_q)
        acc = acc_t.numpy().reshape(-1, 1)
    assert np.array_equal(acc, g["acc"]), f"accuracy differs: {acc.ravel()} vs {g['acc'].ravel()}"


@pytest.mark.parametrize("name", LARGE)
def test_engine_matches_reference_golden_large(name):
    from tclip_amd import engine
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    res = _run(g)
    assert np.array_equal(res.mm_iters.cpu().numpy()[0], g["mm_iters"])
    assert np.array_equal(res.preds.cpu().numpy(), g["argmax"][-1].astype(np.int32))
    alpha = res.alpha.cpu().numpy()
    N = alpha.shape[0]
    rows = g["alpha_rows_idx"]
    sampled = np.stack([alpha[n, rows[n]] for n in range(N)])
    assert _fro_rel(sampled, g["alpha_rows"]).max() <= ALPHA_TOL
    # every row through its float64 checksums: per task in the Frobenius sense (same bar as the
    # small fixtures), per row with the looser bound that single small rows need
    a64 = alpha.astype(np.float64)
    rs, rss = a64.sum(-1), (a64 * a64).sum(-1)
    assert (np.abs(np.sqrt(rss.sum(-1)) / np.sqrt(g["alpha_rowsumsq"].sum(-1)) - 1.0) <= ALPHA_TOL).all()
    np.testing.assert_allclose(rs, g["alpha_rowsum"], rtol=5e-4)
    np.testing.assert_allclose(rss, g["alpha_rowsumsq"], rtol=1e-3)
    assert np.array_equal(res.u.cpu().numpy(), g["u"])
    assert np.array_equal(res.v.cpu().numpy(), g["v"])
    acc_t, _ = engine.clustering_accuracy(torch.from_numpy(g["x_q"]).cuda(), res.preds,
                                          torch.from_numpy(g["y_q"]).squeeze(2))
    assert np.array_equal(acc_t.numpy().reshape(-1, 1), g["acc"])
