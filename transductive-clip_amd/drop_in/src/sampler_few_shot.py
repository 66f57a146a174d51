"""Few-shot task sampling with the reference's random-number consumption
(reference: src/sampler_few_shot.py).  Support: for EVERY class, in class order,
`torch.randperm(len(class))[:shots]`; query: as zero-shot but with the configured k_eff."""
import numpy as np
import torch

from src.sampler_zero_shot import MAX_RESAMPLES


def _indices_by_class(labels, n):
    labels = np.asarray(labels)
    order = np.argsort(labels, kind="stable")
    bounds = np.searchsorted(labels[order], np.arange(n + 1))
    return [torch.from_numpy(order[bounds[c]:bounds[c + 1]]) for c in range(n)]


class CategoriesSampler_few_shot:
    def __init__(self, n_batch, k_eff, n_class, s_shot, n_query, force_query_size=False):
        self.n_batch, self.k_eff, self.s_shot, self.n_query, self.n_class = n_batch, k_eff, s_shot, n_query, n_class
        self.force_query_size = force_query_size
        self.list_classes = list(range(n_class))

    def create_list_classes(self, label_support, label_query):
        n = int(np.max(np.asarray(label_support))) + 1
        self.m_ind_support = _indices_by_class(label_support, n)
        self.m_ind_query = _indices_by_class(label_query, n)


class SamplerSupport_few_shot:
    def __init__(self, cat_samp):
        self.c = cat_samp
        self.n_batch = cat_samp.n_batch

    def __len__(self):
        return self.n_batch

    def __iter__(self):
        c = self.c
        for _ in range(self.n_batch):
            yield torch.cat([c.m_ind_support[j][torch.randperm(len(c.m_ind_support[j]))[:c.s_shot]]
                             for j in c.list_classes])


class SamplerQuery_few_shot:
    def __init__(self, cat_samp):
        self.c = cat_samp
        self.n_batch = cat_samp.n_batch

    def __len__(self):
        return self.n_batch

    def __iter__(self):
        c = self.c
        for _ in range(self.n_batch):
            for attempt in range(MAX_RESAMPLES):
                classes = torch.randperm(len(c.list_classes))[:c.k_eff].tolist()
                pool = torch.cat([c.m_ind_query[j] for j in classes])
                query = pool[torch.randperm(len(pool))[:c.n_query]]
                if len(query) >= c.n_query or not c.force_query_size:
                    break
            else:
                raise RuntimeError("could not draw n_query images from the sampled classes")
            yield query
