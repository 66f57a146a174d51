"""ALPHA_TIM: the torch-autograd oracle against the golden vectors produced by the reference's class
(CPU), and the HIP path (closed-form gradient) against the same vectors (GPU).  The reference's MKL matmuls and
autograd accumulation leave no operation order to reproduce, and Adam normalises every coordinate's step, so the
comparison carries a tolerance.  GPU path: per-fixture bounds at twice the deviation measured on MI355X
(tests/golden/f4_tolerances.json; weights 1e-6 .. 3.5e-4 absolute, query logits 2e-5 .. 2.5e-3, criterions 1e-5 .. 6.4e-4
relative), predictions and accuracies equal.  CPU oracle on a host other than the fixtures': weights to max(2e-4, lr)
(they move by up to iter * lr = 0.1 ... 0.3), logits to 2e-2 of their 1e1..1e2 range, criterions to 1 %."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from oracle import ref_torch

NAMES = golden_names("fs_tim_")


def _params(g):
    return dict(iters=int(g["iters"]), temp=float(g["temp"]), lr=float(g["lr"]), alpha_value=float(g["alpha_value"]),
                loss_weights=[float(w) for w in g["loss_weights"]], entropies=[str(e) for e in g["entropies"]])


def test_fixtures_present():
    assert len(NAMES) >= 6


@pytest.mark.parametrize("name", [n for n in NAMES if "K397" not in n and "K100" not in n])
def test_oracle_reproduces_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if str(g["torch_version"]) != torch.__version__:
        pytest.skip("fixtures were made with another torch build")
    t = ref_torch.run_alpha_tim(torch.from_numpy(g["x_q"]), torch.from_numpy(g["x_s"]), torch.from_numpy(g["y_s"]),
                                n_class=int(g["K"]), **_params(g))
    # bit-identical on the host the fixtures were made on; another CPU takes other MKL kernels and Adam amplifies that
    _check(t["weights"].numpy(), t["logits_q"].numpy(), t["criterions"].numpy(), g)
    acc = (t["argmax"] == torch.from_numpy(g["y_q"]).squeeze(2)).float().mean(1, keepdim=True)
    assert np.abs(acc.numpy() - g["acc"]).max() <= 2 / 75 + 1e-6


GPU_BOUNDS = json.load(open(os.path.join(GOLDEN, "f4_tolerances.json")))["alpha_tim"]


def _check(weights, logits_q, crit, g, bounds=None):
    """bounds=None: the host-to-host tolerance of the CPU oracle; else the fixture's entry of f4_tolerances.json"""
    w_err = np.abs(weights - g["weights"]).max()
    w_tol = bounds["weights_abs"] if bounds else max(2e-4, float(g["lr"]))   # one Adam step of the fixture's learning rate, at least 2e-4
    assert w_err <= w_tol, f"weights differ by {w_err} (bound {w_tol})"
    l_err = np.abs(logits_q - g["logits_q"]).max()
    l_tol = bounds["logits_abs"] if bounds else 2e-2
    assert l_err <= l_tol, f"query logits differ by {l_err} (bound {l_tol})"
    c_err = np.abs(crit / g["criterions"] - 1).max()
    c_tol = bounds["criterions_rel"] if bounds else 1e-2
    assert c_err <= c_tol, f"criterions differ by {c_err} relative (bound {c_tol})"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_reference(name):
    from src.methods.few_shot.tim import ALPHA_TIM
    from src.utils import CfgNode
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K, prm = int(g["K"]), _params(g)
    a = CfgNode(iter=prm["iters"], num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30, shots=int(g["shots"]),
                use_softmax_feature=True, temp=prm["temp"], loss_weights=prm["loss_weights"], lr_alpha_tim=prm["lr"],
                entropies=prm["entropies"], alpha_value=prm["alpha_value"])
    m = ALPHA_TIM(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    logs = m.run_task(task_dic={"x_q": torch.from_numpy(g["x_q"]), "y_q": torch.from_numpy(g["y_q"]),
                                "x_s": torch.from_numpy(g["x_s"]), "y_s": torch.from_numpy(g["y_s"])}, shot=int(g["shots"]))
    assert name in GPU_BOUNDS, "every ALPHA_TIM fixture has its measured bound in tests/golden/f4_tolerances.json"
    _check(m.weights.cpu().numpy(), m.logits_q.cpu().numpy(), logs["criterions"], g, GPU_BOUNDS[name])
    assert logs["criterions"].shape == g["criterions"].shape and logs["acc"].shape == g["acc"].shape
    # everything discrete is equal: every prediction and every accuracy
    assert np.array_equal(m.preds.cpu().numpy(), g["logits_q"].argmax(2)), "predictions differ from the reference's"
    assert np.array_equal(logs["acc"], g["acc"]), "accuracies differ from the reference's"


@pytest.mark.gpu
def test_engine_close_to_oracle_on_fresh_tasks():
    """Seeded inputs no fixture holds, every entropy combination, two batches in one call (the criterion is per batch)."""
    from tclip_amd import engine, synth
    K, N, shots = 21, 4, 2
    x_q, _ = synth.make_query_tasks(N, K, seed=77, k_eff=4)
    x_s, y_s = synth.make_support(N, K, shots, seed=77)
    for ent in (("Shannon", "Alpha", "Alpha"), ("Shannon", "Shannon", "Shannon"), ("Shannon", "Shannon", "Alpha"),
                ("Shannon", "Alpha", "Shannon")):
        prm = dict(iters=150, temp=15.0, lr=1e-3, alpha_value=3.0, loss_weights=[1.0, 0.7, 1.2], entropies=list(ent))
        w, lq, preds, crit = engine.run_alpha_tim(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), n_batches=2, **prm)
        torch.cuda.synchronize()
        for b in range(2):
            sl = slice(b * 2, b * 2 + 2)
            t = ref_torch.run_alpha_tim(x_q[sl], x_s[sl], y_s[sl], n_class=K, **prm)
            assert (w[sl].cpu() - t["weights"]).abs().max() < 5e-4, ent
            assert (lq[sl].cpu() - t["logits_q"]).abs().max() < 5e-2, ent
            torch.testing.assert_close(crit[b].cpu(), t["criterions"], rtol=1e-2, atol=1e-7)
        assert torch.equal(preds.cpu().long(), lq.cpu().argmax(2))


@pytest.mark.gpu
def test_argument_errors():
    from tclip_amd import engine, synth
    x_q, _ = synth.make_query_tasks(2, 6, seed=1, k_eff=3)
    x_s, y_s = synth.make_support(2, 6, 1, seed=1)
    with pytest.raises(ValueError, match="Entropies must be in"):
        engine.run_alpha_tim(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=3, temp=15, lr=1e-4, alpha_value=7.0,
                             entropies=("Shannon", "Renyi", "Alpha"))
    with pytest.raises(RuntimeError, match="alpha_value"):
        engine.run_alpha_tim(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=3, temp=15, lr=1e-4, alpha_value=1.0)
    with pytest.raises(RuntimeError, match="iters"):
        engine.run_alpha_tim(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=0, temp=15, lr=1e-4, alpha_value=7.0)
