"""Task-batch loop of the zero-shot evaluation (reference: src/eval_zero_shot.py:140-187),
rebuilt around the batched engine: the index tensors of ALL batches are drawn first (the method
itself consumes no random numbers, so the reference's RNG stream is reproduced), the feature
table lives on the GPU, rows are gathered there, and every batch of this rank runs in one engine
call with n_batches > 1 (each batch keeps its own MM stop test).  With torch.distributed
initialised, batches are dealt round-robin to the ranks and gathered once onto rank 0.

Feature extraction and dataset handling of the reference's Evaluator are out of scope (they need CLIP
weights and images): run_full_evaluation starts from the saved feature files the reference writes."""
import os

import numpy as np
import torch

from src.methods.zero_shot.em_dirichlet import EM_DIRICHLET
from src.methods.zero_shot.hard_em_dirichlet import HARD_EM_DIRICHLET
from src.methods.zero_shot.em_gaussian import EM_GAUSSIAN
from src.methods.zero_shot.inductive_clip import CLIP
from src.methods.zero_shot.em_gaussian_cov import EM_GAUSSIAN_COV
from src.methods.zero_shot.hard_kmeans import HARD_KMEANS
from src.methods.zero_shot.kl_kmeans import KL_KMEANS
from src.methods.zero_shot.soft_kmeans import SOFT_KMEANS
from src.sampler_zero_shot import CategoriesSampler_zero_shot, SamplerQuery_zero_shot
from src.utils import Logger, compute_confidence_interval
from tclip_amd import engine, features, reporting, sharding

def _as_tensor(x):
    return x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))


_METHODS = {'EM_DIRICHLET': EM_DIRICHLET, 'HARD_EM_DIRICHLET': HARD_EM_DIRICHLET, 'SOFT_KMEANS': SOFT_KMEANS,
            'HARD_KMEANS': HARD_KMEANS, 'EM_GAUSSIAN': EM_GAUSSIAN, 'EM_GAUSSIAN_COV': EM_GAUSSIAN_COV,
            'KL_KMEANS': KL_KMEANS, 'CLIP': CLIP}


class Evaluator_zero_shot:
    def __init__(self, device, args, log_file):
        self.device = device
        self.args = args
        self.log_file = log_file
        self.logger = Logger(__name__, self.log_file)

    def run_full_evaluation(self, model, preprocess):
        """eval_zero_shot.py:44-72 for the case the reference itself short-cuts: when the feature file of the
        split is already saved the reference skips CLIP (utils.py:266-271 "Features already saved ... skipping")
        and goes from the pickle to evaluate_tasks and report_results; `model`/`preprocess` are then unused and
        may be None.  Extracting features needs CLIP and the image datasets and is not part of this package."""
        dic = self.extract_and_load_features(model, None, None)
        all_features_query = dic['concat_features'].to('cpu')
        all_labels_query = dic['concat_labels'].long().to('cpu')
        mean_accuracies, mean_times = self.evaluate_tasks(model, all_features_query, all_labels_query)
        if mean_accuracies is not None:            # rank 0 (every rank without torch.distributed)
            self.report_results(mean_accuracies, mean_times)
        return mean_accuracies, mean_times

    def extract_and_load_features(self, model, dataset, data_loaders):
        """Loads data/<dataset>/saved_features/<split>_{softmax_<backbone>_T<T>,visual_<backbone>}.plk
        (eval_zero_shot.py:89-111) under args.results_root (default: the working directory, as in the reference)."""
        path = reporting.saved_feature_path(self.args, self.args.used_test_set, getattr(self.args, 'results_root', '.'))
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: this package runs the reference's evaluation from saved "
                                    "feature files; extract them with the reference (CLIP is out of scope here)")
        feats, labels = features.load_features(path)
        return {'concat_features': feats, 'concat_labels': labels}

    def report_results(self, mean_accuracies, mean_times):
        """eval_zero_shot.py:189-232."""
        return reporting.report_results(self.args, mean_accuracies, mean_times, self.logger,
                                        root=getattr(self.args, 'results_root', '.'))

    def get_method_builder(self, model, device, args, log_file):
        try:
            cls = _METHODS[args.name_method]
        except KeyError:
            raise ValueError(f"method {args.name_method!r} is not part of the EM-Dirichlet engine")
        return cls(model=model, device=device, log_file=log_file, args=args)

    def sample_indices(self, all_labels_query):
        """(n_batches, batch_size, n_query) int64 index tensor, drawn batch by batch exactly as
        the reference does (a fresh sampler per batch)."""
        a = self.args
        out, lists = [], None
        for _ in range(int(a.number_tasks / a.batch_size)):
            sampler = CategoriesSampler_zero_shot(a.batch_size, a.k_eff, a.n_class, a.n_query, force_query_size=True)
            # the label -> index lists are a function of the labels alone and consume no random numbers: built for the
            # first batch, shared by the others (the reference rebuilds them per batch, sampler_zero_shot.py:31-36)
            if lists is None:
                sampler.create_list_classes(all_labels_query)
                lists = sampler.m_ind_query
            else:
                sampler.m_ind_query = lists
            out.append(torch.stack(list(SamplerQuery_zero_shot(sampler)), 0))
        return torch.stack(out, 0)

    def evaluate_tasks(self, model, all_features_query, all_labels_query, indices=None):
        """`indices`: an index tensor drawn earlier with sample_indices() (bench.py draws it outside its
        timed region); None draws it here, as the reference's loop does.  The feature table and the
        labels may already live on the device."""
        a = self.args
        self.logger.info("=> Runnning evaluation with method {} on {} dataset".format(
            a.name_method, getattr(a, 'used_test_set', 'test')))
        dev = torch.device(self.device)
        table = _as_tensor(all_features_query).float().to(dev)
        labels = _as_tensor(all_labels_query).long()
        idx = self.sample_indices(labels.cpu().numpy()) if indices is None else indices   # every rank: the same stream
        n_batches, N, Q = idx.shape
        mine = sharding.my_batches(n_batches)
        K = table.shape[1]
        method = self.get_method_builder(model=model, device=self.device, args=a, log_file=self.log_file)
        timestamps = 0.0
        if mine:
            my_idx = idx[mine].reshape(-1)
            x_q = engine.gather_rows(table, my_idx).view(len(mine) * N, Q, K)
            y_q = labels[my_idx.to(labels.device)].view(len(mine) * N, Q)
            method.run_method(query=x_q, y_q=y_q.to(dev), n_batches=len(mine))   # SOFT_KMEANS has no cross-task coupling
            logs = method.get_logs()
            parts = sharding.method_parts(a, method, logs, len(mine), N, Q, dev)
            timestamps = float(logs['timestamps'])
        else:      # more ranks than batches: this rank only takes part in the gather
            parts = sharding.method_parts(a, None, None, 0, N, Q, dev)
        # the one collective: per-task predictions, accuracies and the per-batch records of every rank onto rank 0
        got = sharding.gather_packed(parts, n_batches)
        self.last_method = method
        if got is None:                                             # not rank 0
            return None, None
        acc = got['acc'].numpy()
        results_task = [compute_confidence_interval(acc[b])[0] for b in range(n_batches)]
        self.last_task_accuracies = acc
        self.last_task_predictions = got['preds'].view(n_batches, N, Q).numpy()      # class per query, every batch of the run
        self.last_batch_criterions = got['criterions'].numpy() if 'criterions' in got else None     # (n_batches, iter)
        self.last_batch_mm_iters = got['mm_iters'].numpy() if 'mm_iters' in got else None
        return np.asarray(results_task).mean(), timestamps
