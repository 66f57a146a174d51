"""Few-shot task dictionary (reference: src/task_generator_few_shot.py:27-99).  Labels are
re-indexed by `flip(unique(support labels))` and, for softmax features, the feature columns are
permuted the same way, exactly as the reference does."""
import torch


def label_permutation(labels_support):
    """(uniq, lut): new label j stands for old label uniq[j]; lut[old] = new.  Computed on CPU
    labels with the reference's own call, torch.flip(torch.unique(., sorted=False))."""
    uniq = torch.flip(torch.unique(labels_support.cpu(), sorted=False), dims=(0,))
    lut = torch.zeros(int(uniq.max()) + 1, dtype=torch.long)
    lut[uniq] = torch.arange(len(uniq))
    return uniq, lut


def relabel(data_support, data_query, labels_support, labels_query, use_softmax_feature=True):
    if not use_softmax_feature:
        return data_support, data_query, labels_support.long(), labels_query.long()
    uniq, lut = label_permutation(labels_support)
    cols = uniq.to(data_support.device)
    return (data_support[:, cols], data_query[:, cols],
            lut[labels_support.long().cpu()], lut[labels_query.long().cpu()])


class Tasks_Generator_few_shot:
    def __init__(self, k_eff, shot, n_query, n_class, loader_support, loader_query, model, args):
        self.k_eff, self.shot, self.n_query, self.n_class = k_eff, shot, n_query, n_class
        self.loader_support, self.loader_query, self.model, self.args = loader_support, loader_query, model, args

    def generate_tasks(self):
        out = {'x_s': [], 'y_s': [], 'x_q': [], 'y_q': []}
        for (xs, ys), (xq, yq) in zip(self.loader_support, self.loader_query):
            xs, xq, ys, yq = relabel(xs, xq, ys, yq, self.args.use_softmax_feature)
            for k, val in zip(('x_s', 'y_s', 'x_q', 'y_q'), (xs, ys, xq, yq)):
                out[k].append(val)
        return {'x_s': torch.stack(out['x_s'], 0), 'y_s': torch.stack(out['y_s'], 0).unsqueeze(-1),
                'x_q': torch.stack(out['x_q'], 0), 'y_q': torch.stack(out['y_q'], 0).unsqueeze(-1)}
