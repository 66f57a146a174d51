"""Feature files (reference pickle schema) and the probability-feature front-end (F2)."""
import pickle

import numpy as np
import pytest
import torch


def test_pickle_schema_round_trip(tmp_path):
    from tclip_amd import features
    feats, labels = torch.rand(12, 5), torch.arange(12) % 5
    p = str(tmp_path / "test_softmax_RN50_T30.plk")
    features.save_features(p, feats, labels)
    with open(p, "rb") as f:
        d = pickle.load(f)
    assert set(d) == {"concat_features", "concat_labels"}            # src/utils.py:300-306
    f2, l2 = features.load_features(p)
    assert torch.equal(f2, feats) and torch.equal(l2, labels)


@pytest.mark.gpu
@pytest.mark.parametrize("n,D,K,T", [(7, 512, 10, 30), (65, 1024, 100, 30), (33, 512, 1000, 10), (5, 30, 397, 50)])
def test_probability_features_match_formula(n, D, K, T):
    """softmax(T * normalize(f) @ text.T) (src/utils.py:287-290) against an fp64 evaluation of the
    same formula and torch's fp32 one: the reference runs this on its own GPU through cuBLAS, so
    there are no reference bits to match; 2e-6 absolute on probabilities is fp32 GEMM noise."""
    from tclip_amd import features
    g = torch.Generator().manual_seed(n + K)
    f = torch.randn(n, D, generator=g) * 3
    text = torch.randn(K, D, generator=g)
    text /= text.norm(dim=-1, keepdim=True)
    z = features.probability_features(f.cuda(), text.cuda(), T).cpu()
    ref64 = (T * (f.double() / f.double().norm(dim=-1, keepdim=True)) @ text.double().T).softmax(-1)
    fn = f / f.norm(dim=-1, keepdim=True)
    ref32 = (T * fn @ text.T).softmax(-1)
    assert (z.sum(-1) - 1).abs().max() < 1e-5
    assert (z.double() - ref64).abs().max() < 2e-6
    assert (z - ref32).abs().max() < 2e-6
    assert torch.equal(z.argmax(-1), ref64.argmax(-1))
