"""SOFT_KMEANS (BASELINE config 3's second method): the two CPU oracles against the
golden vectors produced by the reference (CPU), and the HIP path against the same vectors (GPU).
Everything is bit-exact: centroids, responsibilities, per-iteration argmax, accuracies."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from oracle import c_oracle, ref_torch

NAMES = golden_names("zs_skm_")


@pytest.mark.parametrize("name", NAMES)
def test_oracles_reproduce_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    c = c_oracle.run_soft_kmeans(g["x_q"], iters=int(g["iters"]), temperature=30)
    assert np.array_equal(c["w"], g["alpha"]) and np.array_equal(c["u"], g["u"])
    assert np.array_equal(c["argmax"], g["argmax"])
    if str(g["torch_version"]) == torch.__version__:
        t = ref_torch.run_soft_kmeans(torch.from_numpy(g["x_q"]), n_class=K, iters=int(g["iters"]), temperature=30)
        assert np.array_equal(t["w"].numpy(), g["alpha"]) and np.array_equal(t["u"].numpy(), g["u"])
        acc, _ = ref_torch.clustering_accuracy(t["u"], torch.from_numpy(g["x_q"]), torch.from_numpy(g["y_q"]).squeeze(2), K)
        assert np.array_equal(acc.numpy(), g["acc"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_reference(name):
    from src.methods.zero_shot.soft_kmeans import SOFT_KMEANS
    from src.utils import CfgNode
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    a = CfgNode(iter=int(g["iters"]), num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30,
                use_softmax_feature=True, graph_matching=True)
    m = SOFT_KMEANS(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    logs = m.run_task(task_dic={"x_q": torch.from_numpy(g["x_q"]), "y_q": torch.from_numpy(g["y_q"])})
    assert np.array_equal(m.w.cpu().numpy(), g["alpha"]), "centroids differ"
    assert np.array_equal(m.u.cpu().numpy(), g["u"]), "responsibilities differ"
    assert np.array_equal(m.preds.cpu().numpy(), g["argmax"][-1].astype(np.int32))
    assert np.array_equal(logs["acc"], g["acc"])
    assert np.array_equal(logs["criterions"], g["criterions"])
