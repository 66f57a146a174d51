"""Stacks sampled (features, labels) pairs into one task dictionary
(reference: src/task_generator_zero_shot.py:36-65): 'x_q' (n_task, n_query, K), 'y_q' (n_task, n_query, 1)."""
import torch


class Tasks_Generator_zero_shot:
    def __init__(self, k_eff, n_query, n_class, loader_query, model, args):
        self.k_eff, self.n_query, self.n_class = k_eff, n_query, n_class
        self.loader_query, self.model, self.args = loader_query, model, args

    def generate_tasks(self):
        xs = [x for x, _ in self.loader_query]
        ys = [y.long() for _, y in self.loader_query]
        return {'x_q': torch.stack(xs, 0), 'y_q': torch.stack(ys, 0).unsqueeze(-1)}
