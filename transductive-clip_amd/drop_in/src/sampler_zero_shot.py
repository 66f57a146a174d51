"""Zero-shot task sampling with the reference's random-number consumption
(reference: src/sampler_zero_shot.py).  Per task: python `random.randint(3, 10)` effective
classes (this overrides args.k_eff, as in the reference), `torch.randperm(K)[:k_eff]` classes,
pool = their image indices concatenated in that order, `pool[torch.randperm(len(pool))[:n_query]]`.
Seeding random/torch as main.py:42-46 does therefore yields the reference's index tensors."""
import random

import numpy as np
import torch

MAX_RESAMPLES = 1000   # the reference retries forever when the drawn classes hold < n_query images


class CategoriesSampler_zero_shot:
    def __init__(self, n_batch, k_eff, n_class, n_query, force_query_size=False):
        self.n_batch, self.k_eff, self.n_class, self.n_query = n_batch, k_eff, n_class, n_query
        self.force_query_size = force_query_size
        self.m_ind_query = []
        self.list_classes = list(range(n_class))

    def create_list_classes(self, label_query):
        label_query = np.asarray(label_query)
        order = np.argsort(label_query, kind="stable")          # ascending indices inside each class
        bounds = np.searchsorted(label_query[order], np.arange(self.n_class + 1))
        self.m_ind_query = [torch.from_numpy(order[bounds[c]:bounds[c + 1]]) for c in range(self.n_class)]


class SamplerQuery_zero_shot:
    def __init__(self, cat_samp):
        self.c = cat_samp
        self.n_batch = cat_samp.n_batch

    def __len__(self):
        return self.n_batch

    def __iter__(self):
        c = self.c
        for _ in range(self.n_batch):
            k_eff = random.randint(3, 10)
            for attempt in range(MAX_RESAMPLES):
                classes = torch.randperm(c.n_class)[:k_eff].tolist()
                pool = torch.cat([c.m_ind_query[j] for j in classes])
                query = pool[torch.randperm(len(pool))[:c.n_query]]
                if len(query) >= c.n_query or not c.force_query_size:
                    break
            else:
                raise RuntimeError("could not draw n_query images from the sampled classes")
            yield query
