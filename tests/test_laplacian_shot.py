"""LAPLACIAN_SHOT: the numpy oracle against golden vectors produced by the reference's class (CPU, with
the removed alias `np.float` restored for that process - tests/golden/make_golden_lshot.py), and the HIP path against
the same vectors (GPU).  The reference's host code (numpy pairwise sums and exp, sklearn's distance kernels) is not
reproduced bit for bit: the neighbour lists, every per-update accuracy and the final assignment must be EQUAL, the unary
term and the bound energies agree within per-fixture bounds at twice the deviation measured on MI355X
(tests/golden/f4_tolerances.json: unary <= 1.3e-6 relative, energies <= 2.3e-7 relative)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from oracle import ref_torch

NAMES = golden_names("fs_lshot_")
BOUNDS = json.load(open(os.path.join(GOLDEN, "f4_tolerances.json")))["laplacian_shot"]


def test_fixtures_present():
    assert len(NAMES) >= 6


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    t = ref_torch.run_laplacian_shot(torch.from_numpy(g["x_q"]), torch.from_numpy(g["x_s"]), torch.from_numpy(g["y_s"]),
                                     torch.from_numpy(g["y_q"]), n_class=int(g["K"]), iters=int(g["iters"]), knn=int(g["knn"]),
                                     lmd=float(g["lmd"]), norm_type=str(g["norm_type"]))
    assert np.array_equal(t["neighbours"], g["neighbours"]) and np.array_equal(t["preds"], g["preds"])
    assert np.allclose(t["unary"], g["unary"], rtol=1e-6, atol=0) and np.array_equal(t["acc"].astype(np.float32), g["acc"])
    assert np.allclose(t["ent_energy"], g["ent_energy"], rtol=1e-9, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_reference(name):
    from src.methods.few_shot.laplacian_shot import LAPLACIAN_SHOT
    from src.utils import CfgNode
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K, N = int(g["K"]), int(g["N"])
    a = CfgNode(iter=int(g["iters"]), num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30, shots=int(g["shots"]),
                use_softmax_feature=True, knn=int(g["knn"]), lmd=float(g["lmd"]), norm_type=str(g["norm_type"]), temp=30,
                batch_size=N)
    m = LAPLACIAN_SHOT(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    logs = m.run_task(task_dic={"x_q": torch.from_numpy(g["x_q"]), "y_q": torch.from_numpy(g["y_q"]),
                                "x_s": torch.from_numpy(g["x_s"]), "y_s": torch.from_numpy(g["y_s"])}, shot=int(g["shots"]))
    assert np.array_equal(np.sort(m.neighbours.cpu().numpy(), axis=2), g["neighbours"]), "kNN graph differs"
    b = BOUNDS[name]                                   # twice the deviation measured on MI355X (tests/golden/f4_tolerances.json)
    du = np.abs(m.unary.cpu().numpy() - g["unary"]) / np.maximum(np.abs(g["unary"]), 1e-30)
    assert du.max() <= b["unary_rel"], f"unary term differs by {du.max():.2e} relative (bound {b['unary_rel']:.1e})"
    assert np.array_equal(m.preds.cpu().numpy(), g["preds"]), "final assignment differs"
    assert logs["acc"].shape == g["acc"].shape and np.array_equal(logs["acc"], g["acc"]), "per-update accuracies differ"
    assert logs["ent_energy"].shape == g["ent_energy"].shape
    de = np.abs(np.asarray(logs["ent_energy"]) / g["ent_energy"] - 1).max()
    assert de <= b["energy_rel"], f"energies differ by {de:.2e} relative (bound {b['energy_rel']:.1e})"
    assert logs["criterions"] == [[0]] * N


@pytest.mark.gpu
def test_engine_equals_oracle_on_fresh_tasks():
    from tclip_amd import engine, synth
    K, N, shots = 21, 6, 2
    x_q, y_q = synth.make_query_tasks(N, K, seed=91, k_eff=4)
    x_s, y_s = synth.make_support(N, K, shots, seed=91)
    for knn, lmd, norm in ((3, 0.7, "L2N"), (6, 2.0, "UN"), (2, 0.1, "L2N")):
        unary, nbr, preds_iter, e = engine.run_laplacian_shot(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=15, knn=knn,
                                                              lmd=lmd, norm_type=norm)
        torch.cuda.synchronize()
        t = ref_torch.run_laplacian_shot(x_q, x_s, y_s, y_q, n_class=K, iters=15, knn=knn, lmd=lmd, norm_type=norm)
        assert np.array_equal(np.sort(nbr.cpu().numpy(), axis=2), t["neighbours"])
        assert np.array_equal(preds_iter[:, -1].cpu().numpy(), t["preds"])
        assert np.allclose(e.cpu().numpy(), t["ent_energy"], rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_argument_errors():
    from tclip_amd import engine, synth
    x_q, _ = synth.make_query_tasks(2, 6, seed=1, k_eff=3)
    x_s, y_s = synth.make_support(2, 6, 1, seed=1)
    with pytest.raises(ValueError, match="norm_type"):
        engine.run_laplacian_shot(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=3, knn=3, lmd=0.7, norm_type="CL2N")
    with pytest.raises(RuntimeError, match="knn"):
        engine.run_laplacian_shot(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=3, knn=1, lmd=0.7)
    with pytest.raises(RuntimeError, match="iters"):
        engine.run_laplacian_shot(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=0, knn=3, lmd=0.7)


@pytest.mark.gpu
def test_nan_features_stay_in_bounds():
    """A zero query row has no L2 norm: its distances are NaN in the reference too.  The call must come back (neighbour
    lists inside the task, assignments inside 0..K-1) and leave the other tasks untouched."""
    from tclip_amd import engine, synth
    K, N = 12, 3
    x_q, _ = synth.make_query_tasks(N, K, seed=5, k_eff=4)
    x_s, y_s = synth.make_support(N, K, 2, seed=5)
    clean = engine.run_laplacian_shot(x_q.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=10, knn=3, lmd=0.7)
    x_bad = x_q.clone()
    x_bad[1, 7] = 0.0
    unary, nbr, preds_iter, e = engine.run_laplacian_shot(x_bad.cuda(), x_s.cuda(), y_s.squeeze(2).cuda(), iters=10, knn=3, lmd=0.7)
    torch.cuda.synchronize()
    assert int(nbr.min()) >= 0 and int(nbr.max()) < 75 and int(preds_iter.min()) >= 0 and int(preds_iter.max()) < K
    for k in (0, 2):
        assert torch.equal(preds_iter[k], clean[2][k]) and torch.equal(e[k], clean[3][k])
