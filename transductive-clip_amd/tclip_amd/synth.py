"""Seeded synthetic probability features for parity runs and the bench.

The reference ships no saved features (``data/`` and ``*.plk`` are git-ignored), so every
input here is synthetic, shaped like the CLIP "probability features"
``z = softmax(T * cos(img, text_k))`` of the reference (README.md:44-47,
src/utils.py:287-290): peaked rows on the K-simplex, 75 queries drawn from 3..10 classes.
Recipe fixed by SURVEY.md section 8(d) so that the GPU path, the oracle and the golden
vectors all see the same tensors for a given seed.
"""
import torch


def _peaked_rows(labels, n_class, gen, boost=3.0, temp=3.0):
    n = labels.shape[0]
    logits = torch.randn(n, n_class, generator=gen)
    logits[torch.arange(n), labels] += boost
    return torch.softmax(temp * logits, dim=-1)


def make_query_tasks(n_task, n_class, seed=2020, n_query=75, k_eff=None):
    """Returns ``x_q (n_task, n_query, n_class) f32`` and ``y_q (n_task, n_query, 1) i64``.

    ``k_eff=None`` draws the number of represented classes uniformly from 3..10 per task
    (what src/sampler_zero_shot.py:52-54 does); an int fixes it (few-shot sampler,
    src/sampler_few_shot.py:92-100).
    """
    gen = torch.Generator().manual_seed(seed)
    x_q = torch.empty(n_task, n_query, n_class)
    y_q = torch.empty(n_task, n_query, 1, dtype=torch.int64)
    for n in range(n_task):
        if k_eff is None:
            ke = int(torch.randint(3, 11, (1,), generator=gen))
        else:
            ke = int(k_eff)
        ke = min(ke, n_class)
        classes = torch.randperm(n_class, generator=gen)[:ke]
        y = classes[torch.randint(ke, (n_query,), generator=gen)]
        x_q[n] = _peaked_rows(y, n_class, gen)
        y_q[n, :, 0] = y
    return x_q, y_q


def make_support(n_task, n_class, shots, seed=2020):
    """Class-sorted support set: ``shots`` rows for every class (src/sampler_few_shot.py:64-76).

    Returns ``x_s (n_task, n_class*shots, n_class) f32`` and ``y_s (n_task, n_class*shots, 1) i64``.
    """
    gen = torch.Generator().manual_seed(seed + 7919)
    y = torch.arange(n_class).repeat_interleave(shots)
    x_s = torch.empty(n_task, n_class * shots, n_class)
    for n in range(n_task):
        x_s[n] = _peaked_rows(y, n_class, gen)
    y_s = y.view(1, -1, 1).repeat(n_task, 1, 1).contiguous()
    return x_s, y_s


def make_feature_table(n_class, rows_per_class, seed=2020):
    """Class-sorted ``(n_class*rows_per_class, n_class)`` feature table + labels, the schema
    of the reference's ``{'concat_features','concat_labels'}`` pickles (src/utils.py:300-306)."""
    gen = torch.Generator().manual_seed(seed + 104729)
    labels = torch.arange(n_class).repeat_interleave(rows_per_class)
    feats = _peaked_rows(labels, n_class, gen)
    return feats, labels
