"""Saved-feature files and the probability-feature front-end.

The reference stores features as pickles `{'concat_features': (n, K) f32 tensor, 'concat_labels':
(n,) tensor}` under data/<dataset>/saved_features/<split>_softmax_<backbone>_T<T>.plk
(src/utils.py:266-267, 300-306; visual embeddings: `<split>_visual_<backbone>.plk`, :324-325,
:343-360).  load_features / save_features read and write that schema, so features extracted with
the reference flow into evaluate_tasks unchanged.  probability_features turns L2-normalisable
visual embeddings into the softmax features EM-Dirichlet needs, on the GPU."""
import ctypes
import pickle

import torch

from . import _capi
from .engine import _ptr, _require_cuda, _stream


def load_features(path):
    with open(path, "rb") as f:
        d = pickle.load(f)
    feats = torch.as_tensor(d["concat_features"]).float()
    labels = torch.as_tensor(d["concat_labels"]).long()
    return feats, labels


def save_features(path, features, labels):
    with open(path, "wb") as f:
        pickle.dump({"concat_features": torch.as_tensor(features).float().cpu(),
                     "concat_labels": torch.as_tensor(labels).cpu()}, f)


def probability_features(visual, text_features, temperature):
    """visual (n, D) cuda f32, text_features (K, D) cuda f32 with unit-norm rows -> (n, K) cuda f32:
    softmax(T * normalize(visual) @ text_features.T), the formula of src/utils.py:287-290."""
    _require_cuda(visual, "visual")
    visual = visual.contiguous().float()
    text = text_features.to(visual.device).contiguous().float()
    n, D = visual.shape
    K = text.shape[0]
    out = torch.empty(n, K, device=visual.device)
    with torch.cuda.device(visual.device):
        rc = _capi.lib().tclip_probability_features(_ptr(visual), _ptr(text), n, D, K, ctypes.c_float(float(temperature)),
                                                    _ptr(out), _stream())
    _capi.check(rc, "tclip_probability_features")
    return out
