"""EM_GAUSSIAN_COV: the torch-eager oracle against the golden vectors produced by
the reference (CPU), and the HIP path against the same vectors (GPU).  Everything is bit-exact:
centroids, inverse covariances, responsibilities, v, per-iteration argmax, accuracies."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from oracle import ref_torch

NAMES = golden_names("zs_emgc_")


def test_fixtures_present():
    assert len(NAMES) >= 6


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if str(g["torch_version"]) != torch.__version__:
        pytest.skip("fixtures were made with another torch build")
    K = int(g["K"])
    t = ref_torch.run_em_gaussian_cov(torch.from_numpy(g["x_q"]), n_class=K, iters=int(g["iters"]), lambd=int(K / 5) * 75)
    assert np.array_equal(t["w"].numpy(), g["alpha"]) and np.array_equal(t["u"].numpy(), g["u"])
    assert np.array_equal(t["s"].numpy(), g["s"]) and np.array_equal(t["v"].numpy(), g["v"])
    assert np.array_equal(t["argmax"].numpy().astype(np.int16), g["argmax"])
    assert (g["criterions"] == 0).all()
    acc, _ = ref_torch.clustering_accuracy(t["u"], torch.from_numpy(g["x_q"]), torch.from_numpy(g["y_q"]).squeeze(2), K)
    assert np.array_equal(acc.numpy(), g["acc"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_reference(name):
    from src.methods.zero_shot.em_gaussian_cov import EM_GAUSSIAN_COV
    from src.utils import CfgNode
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    a = CfgNode(iter=int(g["iters"]), num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30,
                use_softmax_feature=True, graph_matching=True)
    m = EM_GAUSSIAN_COV(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    logs = m.run_task(task_dic={"x_q": torch.from_numpy(g["x_q"]), "y_q": torch.from_numpy(g["y_q"])})
    assert np.array_equal(m.w.cpu().numpy(), g["alpha"]), "centroids differ"
    assert np.array_equal(m.s.cpu().numpy(), g["s"]), "inverse covariances differ"
    assert np.array_equal(m.u.cpu().numpy(), g["u"]), "responsibilities differ"
    assert np.array_equal(m.v.cpu().numpy(), g["v"]), "v differs"
    assert np.array_equal(m.preds.cpu().numpy(), g["argmax"][-1].astype(np.int32))
    assert np.array_equal(logs["acc"], g["acc"])
    assert np.array_equal(logs["criterions"], g["criterions"])
