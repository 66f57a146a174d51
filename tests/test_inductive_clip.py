"""Inductive CLIP baseline (reference src/methods/zero_shot/inductive_clip.py): the arg-max kernel
against torch.argmax (first maximum on ties, NaN counts as the maximum), and the method class."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1, 2, 7, 37, 100, 397, 1000])
def test_argmax_rows_matches_torch(K):
    from tclip_amd import engine
    g = torch.Generator().manual_seed(K)
    x = torch.rand(3, 75, K, generator=g)
    x[0, 0, :] = 0.25                               # all tied: first index
    if K > 2:
        x[0, 1, K // 2] = x[0, 1, K - 1] = 2.0      # two maxima
        x[1, 3, K // 3] = float("nan")              # torch: a NaN is the maximum
        x[1, 4, 1] = x[1, 4, K - 1] = float("nan")  # first NaN
    got = engine.argmax_rows(x.cuda()).cpu().long()
    assert torch.equal(got, x.argmax(2))


def test_method_class_matches_reference_semantics():
    from src.methods.zero_shot.inductive_clip import CLIP
    from src.utils import CfgNode
    from tclip_amd import synth
    x_q, y_q = synth.make_query_tasks(6, 37, seed=5)
    a = CfgNode(use_softmax_feature=True, num_classes_test=37, n_class=37)
    m = CLIP(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    logs = m.run_task(task_dic={"x_q": x_q, "y_q": y_q})
    want = (x_q.argmax(2) == y_q.squeeze(2)).float().mean(1, keepdim=True).numpy()
    assert np.array_equal(logs["acc"], want)
    assert logs["criterions"].shape == (1,) and logs["criterions"][0] == 0 and logs["timestamps"] == 0
