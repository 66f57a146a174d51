"""Task-batch loop of the few-shot evaluation (reference: src/eval_few_shot.py:213-270) on the
batched engine; see src/eval_zero_shot.py for the structure.  Per batch the reference draws all
query index sets first, then all support index sets (eval_few_shot.py:233-241); the same order is
kept so that a seeded run sees the reference's tasks."""
import os

import numpy as np
import torch

from src.methods.few_shot.em_dirichlet import EM_DIRICHLET
from src.methods.few_shot.hard_em_dirichlet import HARD_EM_DIRICHLET
from src.methods.few_shot.paddle import PADDLE
from src.methods.few_shot.bdcspn import BDCSPN
from src.methods.few_shot.tim import ALPHA_TIM
from src.methods.few_shot.laplacian_shot import LAPLACIAN_SHOT
from src.sampler_few_shot import CategoriesSampler_few_shot, SamplerQuery_few_shot, SamplerSupport_few_shot
from src.task_generator_few_shot import label_permutation, relabel
from src.utils import Logger, compute_confidence_interval
from tclip_amd import engine, features, reporting, sharding

def _as_tensor(x):
    return x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))


def relabel_batch(x_s, x_q, y_s, y_q, use_softmax_feature):
    xs2, xq2, ys2, yq2 = [], [], [], []
    for t in range(x_s.shape[0]):
        a_, b_, c_, d_ = relabel(x_s[t], x_q[t], y_s[t], y_q[t], use_softmax_feature)
        xs2.append(a_)
        xq2.append(b_)
        ys2.append(c_)
        yq2.append(d_)
    return torch.stack(xs2, 0), torch.stack(xq2, 0), torch.stack(ys2, 0), torch.stack(yq2, 0)


def relabel_indices(y_s, y_q, n_class):
    """Tasks_Generator_few_shot.get_task (task_generator_few_shot.py:41-52) for a whole set of tasks WITHOUT touching the
    features: per task the column permutation `unique_labels` as an int32 row and the re-indexed support / query labels.
    None when some task's support set misses a class (the permuted task would have fewer than n_class columns; the caller
    materialises the tensors then, as the reference does)."""
    y_s, y_q = y_s.long(), y_q.long()
    if y_s.numel() and int(y_s.min()) >= 0 and int(y_s.max()) < n_class:
        present = torch.zeros(y_s.shape[0], n_class, dtype=torch.bool).scatter_(1, y_s, True)
        if bool(present.all()):
            # every class in every support set (what the reference's sampler draws): unique_labels = flip(arange(K)) for
            # every task, i.e. column d <- table column K-1-d and label y <- K-1-y, without a torch.unique per task
            cols = torch.arange(n_class - 1, -1, -1, dtype=torch.int32).repeat(y_s.shape[0], 1)
            return cols, n_class - 1 - y_s, n_class - 1 - y_q
    cols, ys2, yq2 = [], [], []
    for t in range(y_s.shape[0]):
        uniq, lut = label_permutation(y_s[t])
        if len(uniq) != n_class:
            return None
        cols.append(uniq.to(torch.int32))
        ys2.append(lut[y_s[t].long()])
        yq2.append(lut[y_q[t].long()])
    return torch.stack(cols, 0), torch.stack(ys2, 0), torch.stack(yq2, 0)


_METHODS = {'EM_DIRICHLET': EM_DIRICHLET, 'HARD_EM_DIRICHLET': HARD_EM_DIRICHLET, 'PADDLE': PADDLE, 'BDCSPN': BDCSPN,
            'ALPHA_TIM': ALPHA_TIM, 'LAPLACIAN_SHOT': LAPLACIAN_SHOT}


class Evaluator_few_shot:
    def __init__(self, device, args, log_file):
        self.device = device
        self.args = args
        self.log_file = log_file
        self.logger = Logger(__name__, self.log_file)

    def run_full_evaluation(self, model, preprocess):
        """eval_few_shot.py:42-74 from saved feature files (see Evaluator_zero_shot.run_full_evaluation): the
        support table is the train split, the query table the `used_test_set` split."""
        dic_s, dic_q = self.extract_and_load_features(model, None, None)
        mean_accuracies, mean_times = self.evaluate_tasks(
            model, dic_s['concat_features'].to('cpu'), dic_s['concat_labels'].long().to('cpu'),
            dic_q['concat_features'].to('cpu'), dic_q['concat_labels'].long().to('cpu'))
        if mean_accuracies is not None:
            self.report_results(mean_accuracies, mean_times)
        return mean_accuracies, mean_times

    def extract_and_load_features(self, model, dataset, data_loaders):
        """eval_few_shot.py:91-128, loading only."""
        root = getattr(self.args, 'results_root', '.')
        out = []
        for split in ('train', self.args.used_test_set):
            path = reporting.saved_feature_path(self.args, split, root)
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: this package runs the reference's evaluation from saved "
                                        "feature files; extract them with the reference (CLIP is out of scope here)")
            feats, labels = features.load_features(path)
            out.append({'concat_features': feats, 'concat_labels': labels})
        return out[0], out[1]

    def report_results(self, mean_accuracies, mean_times):
        """eval_few_shot.py:272-338 (validation sweep row, or the test-mode row)."""
        return reporting.report_results(self.args, mean_accuracies, mean_times, self.logger,
                                        root=getattr(self.args, 'results_root', '.'))

    def get_method_builder(self, model, device, args, log_file):
        try:
            cls = _METHODS[args.name_method]
        except KeyError:
            raise ValueError(f"method {args.name_method!r} is not part of the EM-Dirichlet engine")
        return cls(model=model, device=device, log_file=log_file, args=args)

    # ---- validation-tuned parameter (reference: eval_few_shot.py:130-187)
    _TUNED = {'LAPLACIAN_SHOT': 'lmd', 'ALPHA_TIM': 'alpha_value', 'PADDLE': 'lambd', 'BDCSPN': 'temp'}

    def set_value_opt_param(self, opt_param):
        if self.args.name_method in self._TUNED:
            self.args[self._TUNED[self.args.name_method]] = opt_param

    def set_method_opt_param(self):
        """Test runs of a tunable method use the parameter that did best on the validation split: the
        sweep file results_few_shot/val/<dataset>/<METHOD>_<softmax|visual>_s<shots>.txt (imagenet borrows
        caltech101's, eval_few_shot.py:161-166), rows `param<TAB>acc`, the LAST best row wins.  A missing or
        unreadable file is an error, as in the reference."""
        a = self.args
        word = '_softmax' if a.use_softmax_feature else '_visual'
        dataset = 'caltech101' if a.dataset == 'imagenet' else a.dataset
        name_file = os.path.join(getattr(a, 'results_root', '.'), 'results_few_shot', 'val', str(dataset),
                                 '{}_s{}.txt'.format(a.name_method + word, a.shots))
        try:
            params, accs = [], []
            with open(name_file, 'r') as f:
                for i, line in enumerate(f):
                    if i < 2:
                        continue
                    cells = line.split('\t')
                    params.append(float(cells[0]))
                    accs.append(float(cells[1]))
            accs = np.array(accs)
            opt_param = params[int(np.argwhere(accs == np.amax(accs))[-1][0])]
        except Exception:
            raise ValueError("The optimal parameter was not found. Please make sure you have performed the "
                             "tuning of the parameter on the validation set.")
        self.logger.info('opt param {}'.format(opt_param))
        self.set_value_opt_param(opt_param)
        return opt_param

    def sample_indices(self, all_labels_support, all_labels_query):
        a = self.args
        q_all, s_all, lists = [], [], None
        for _ in range(int(a.number_tasks / a.batch_size)):
            sampler = CategoriesSampler_few_shot(a.batch_size, a.k_eff, a.n_class, a.shots, a.n_query,
                                                 force_query_size=True)
            if lists is None:             # label -> index lists: no random numbers involved, built once (see eval_zero_shot.py)
                sampler.create_list_classes(all_labels_support, all_labels_query)
                lists = (sampler.m_ind_support, sampler.m_ind_query)
            else:
                sampler.m_ind_support, sampler.m_ind_query = lists
            q_all.append(torch.stack(list(SamplerQuery_few_shot(sampler)), 0))
            s_all.append(torch.stack(list(SamplerSupport_few_shot(sampler)), 0))
        return torch.stack(s_all, 0), torch.stack(q_all, 0)

    def evaluate_tasks(self, model, all_features_support, all_labels_support, all_features_query, all_labels_query,
                       indices=None):
        """`indices`: a (support, query) index pair drawn earlier with sample_indices(); None draws it here."""
        a = self.args
        self.logger.info("=> Runnning evaluation with method {} on {} dataset".format(
            a.name_method, getattr(a, 'used_test_set', 'test')))
        dev = torch.device(self.device)
        tab_s = _as_tensor(all_features_support).float().to(dev)
        tab_q = _as_tensor(all_features_query).float().to(dev)
        lab_s = _as_tensor(all_labels_support).long().cpu()
        lab_q = _as_tensor(all_labels_query).long().cpu()
        s_idx, q_idx = self.sample_indices(lab_s.numpy(), lab_q.numpy()) if indices is None else indices
        n_batches, N, S = s_idx.shape
        Q = q_idx.shape[2]
        mine = sharding.my_batches(n_batches)
        K = tab_q.shape[1]
        # The parameter tuned on the validation split, when the test split is evaluated.  The reference builds the
        # method of a batch FIRST and reads the sweep file afterwards (eval_few_shot.py:250-254), and every method
        # copies its parameter in __init__ (paddle.py:26, bdcspn.py:18, tim.py:198, laplacian_shot.py:32): its batch 0
        # therefore runs with the YAML default and only batches 1.. with the tuned value.  That is reproduced here, so
        # that result files agree with the reference's; `tuned_param_for_every_batch: True` applies it to batch 0 too.
        tuned = getattr(a, 'used_test_set', 'test') == 'test' and getattr(a, 'tunable', False) \
            and not getattr(a, 'skip_tuned_param', False)
        first_method = None
        if tuned:
            if 0 in mine and not getattr(a, 'tuned_param_for_every_batch', False):
                first_method = self.get_method_builder(model=model, device=self.device, args=a, log_file=self.log_file)
            self.set_method_opt_param()
        method = self.get_method_builder(model=model, device=self.device, args=a, log_file=self.log_file)
        timestamps, parts = [], None
        runs = [(first_method, mine[:1]), (method, mine[1:])] if first_method is not None else [(method, mine)]
        ran = None                        # the last method object that actually executed on this rank
        for m, ids in runs:
            if not ids:
                continue
            ran = m
            si, qi = s_idx[ids].reshape(-1), q_idx[ids].reshape(-1)
            # The EM-Dirichlet classes read the task rows from the tables through the index tensors (label flip and column
            # permutation inside the kernels): x_s (T,S,K) - 1.6 GB per 100 tasks at K = 1000, 4 shots - is never built
            if a.name_method in ('EM_DIRICHLET', 'HARD_EM_DIRICHLET') and a.use_softmax_feature \
                    and not getattr(a, 'materialise_tasks', False):
                rel = relabel_indices(lab_s[si].view(-1, S), lab_q[qi].view(-1, Q), K)
                if rel is not None:
                    cols, y_s, y_q = rel
                    m.run_tables(table_s=tab_s, s_idx=si.view(-1, S), table_q=tab_q, q_idx=qi.view(-1, Q), cols=cols,
                                 y_s=y_s.to(dev), y_q=y_q.to(dev), n_batches=len(ids))
                    logs = m.get_logs()
                    parts = sharding.concat_parts(parts, sharding.method_parts(a, m, logs, len(ids), N, Q, dev))
                    timestamps += [float(logs['timestamps'])] * len(ids)
                    continue
            x_s = engine.gather_rows(tab_s, si).view(len(ids) * N, S, K)
            x_q = engine.gather_rows(tab_q, qi).view(len(ids) * N, Q, K)
            y_s, y_q = lab_s[si].view(-1, S), lab_q[qi].view(-1, Q)
            # label re-indexing / column permutation of Tasks_Generator_few_shot.get_task, per task
            x_s, x_q, y_s, y_q = relabel_batch(x_s, x_q, y_s, y_q, a.use_softmax_feature)
            # BDCSPN normalises the features in run_task, before run_method (few_shot/bdcspn.py:165-166): run_batch does both
            run = getattr(m, "run_batch", m.run_method)
            run(support=x_s, query=x_q, y_s=y_s.to(dev), y_q=y_q.to(dev), n_batches=len(ids))
            logs = m.get_logs()
            parts = sharding.concat_parts(parts, sharding.method_parts(a, m, logs, len(ids), N, Q, dev))
            timestamps += [float(logs['timestamps'])] * len(ids)
        if parts is None:      # more ranks than batches: this rank only takes part in the gather
            parts = sharding.method_parts(a, None, None, 0, N, Q, dev)
        got = sharding.gather_packed(parts, n_batches)
        # a rank whose only batch is batch 0 of a tuned run executes `first_method` alone: `method` never ran there and
        # carries no records (mm_iters None); a rank without a batch keeps the freshly built object
        self.last_method = ran if ran is not None else method
        if got is None:
            return None, None
        acc = got['acc'].numpy()
        results_task = [compute_confidence_interval(acc[b])[0] for b in range(n_batches)]
        self.last_task_accuracies = acc
        self.last_task_predictions = got['preds'].view(n_batches, N, Q).numpy()
        self.last_batch_criterions = got['criterions'].numpy() if 'criterions' in got else None
        self.last_batch_mm_iters = got['mm_iters'].numpy() if 'mm_iters' in got else None
        return np.asarray(results_task).mean(), (float(np.mean(timestamps)) if timestamps else 0.0)
