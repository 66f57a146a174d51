"""Shared host logic of the four EM-Dirichlet method classes.

Mirrors the reference's BASE/EM_DIRICHLET interface (constructor keywords, run_task, the logs
dict, the post-call attributes u / v / alpha) while the loop itself runs in libtclip.so on the
GPU (tclip_amd.engine).  Reference: src/methods/zero_shot/em_dirichlet.py:9-246,
src/methods/few_shot/em_dirichlet.py:9-220 and their hard_ twins."""
import time

import numpy as np
import torch

from src.utils import Logger
from tclip_amd import engine

_SIMPLEX_ERROR = "The selected method is unable to handle query features that are not in the unit simplex"


class EMDirichletBase(object):
    HARD = False
    FEW_SHOT = False
    BANNER = "EM-DIRICHLET"

    def __init__(self, model, device, log_file, args):
        self.device = device
        self.iter = args.iter
        # `lambd` from the YAML is ignored by the reference too (em_dirichlet.py:14)
        if self.FEW_SHOT:
            self.lambd = int(args.num_classes_test / args.k_eff) * args.n_query
        else:
            self.lambd = int(args.num_classes_test / 5) * args.n_query
        self.model = model
        self.log_file = log_file
        self.logger = Logger(__name__, self.log_file)
        self.init_info_lists()
        self.args = args
        self.eps = 1e-15
        self.iter_mm = args.iter_mm
        self.mm_iters = None

    def __del__(self):
        try:
            self.logger.del_logger()
        except Exception:
            pass

    def init_info_lists(self):
        self.timestamps = []
        self.criterions = []
        self.test_acc = []

    def get_logs(self):
        self.criterions = np.asarray(self.criterions, dtype=np.float32)
        self.test_acc = torch.cat(self.test_acc, dim=1).cpu().numpy()
        return {'timestamps': np.array(self.timestamps).mean(), 'criterions': self.criterions,
                'acc': self.test_acc}

    # -- accuracy ---------------------------------------------------------------------------
    def compute_acc(self, y_q):
        # on the host: the mean of 75 zeros and ones is rounded as the reference's CPU op rounds it
        preds_q = self.preds.long().cpu()
        accuracy = (preds_q == y_q.cpu()).float().mean(1, keepdim=True)
        self.test_acc.append(accuracy)

    def compute_acc_clustering(self, query, y_q):
        acc, new_preds = engine.clustering_accuracy(query, self.preds, y_q,
                                                    graph_matching=bool(self.args.graph_matching))
        self.matched_preds = new_preds
        self.test_acc.append(acc.view(-1, 1))

    # -- the loop ---------------------------------------------------------------------------
    def _run_engine(self, query, support=None, y_s=None, n_batches=1, tables=None):
        """`tables` = dict(table_q, q_idx[, table_s, s_idx, cols]): the task rows are read from the feature tables through the
        task-batch loop's index tensors (engine.run_em_dirichlet_tasks) and `query` / `support` are not used"""
        if not self.args.use_softmax_feature:
            raise ValueError(_SIMPLEX_ERROR)
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("EM-Dirichlet on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing {} with LAMBDA = {} and T = {}".format(self.BANNER, self.lambd, self.args.T))
        n_task = tables["q_idx"].shape[0] if tables is not None else query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        kw = dict(n_batches=n_batches, iters=self.iter, iter_mm=self.iter_mm, lambd=self.lambd, hard=self.HARD)
        if tables is not None:
            res = engine.run_em_dirichlet_tasks(tables["table_q"], tables["q_idx"], tables.get("table_s"), tables.get("s_idx"),
                                                y_s, tables.get("cols"), **kw)
        else:
            res = engine.run_em_dirichlet(query, support, y_s, **kw)
        torch.cuda.synchronize(dev)
        total = time.time() - t0
        self.u, self.v, self.alpha, self.preds = res.u, res.v, res.alpha, res.preds
        self.mm_iters = res.mm_iters.cpu().numpy()
        crit = res.criterions.cpu().numpy()            # (n_batches, iters)
        # the reference appends one cumulative wall time per outer iteration, divided by n_task
        # (em_dirichlet.py:242-244); the fused loop has no per-iteration host clock, so the total
        # is spread evenly, which reproduces the reference's "mean of cumulative times" statistic
        for i in range(self.iter):
            self.timestamps.append(total * (i + 1) / max(self.iter, 1) / n_task)
        self.criterions = list(crit.mean(0)) if n_batches > 1 else list(crit[0])
        self.criterions_per_batch = crit
        return res


class ZeroShotMixin:
    def run_task(self, task_dic):
        y_q = task_dic['y_q']
        query = task_dic['x_q']
        query = query.to(self.device).float()
        y_q = y_q.long().squeeze(2).to(self.device)
        del task_dic
        self.run_method(query=query, y_q=y_q)
        return self.get_logs()

    def run_method(self, query, y_q, n_batches=1):
        self._run_engine(query, n_batches=n_batches)
        self.compute_acc_clustering(query, y_q)


class FewShotMixin:
    def run_task(self, task_dic, shot=10):
        y_s, y_q = task_dic['y_s'], task_dic['y_q']
        support, query = task_dic['x_s'], task_dic['x_q']
        support = support.to(self.device).float()
        query = query.to(self.device).float()
        y_s = y_s.long().squeeze(2).to(self.device)
        y_q = y_q.long().squeeze(2).to(self.device)
        del task_dic
        self.run_method(support=support, query=query, y_s=y_s, y_q=y_q)
        return self.get_logs()

    def run_method(self, support, query, y_s, y_q, n_batches=1):
        # unlike the reference (few_shot/em_dirichlet.py:186-190) the inputs are left untouched:
        # the engine keeps its own log-features
        self._run_engine(query, support, y_s, n_batches=n_batches)
        self.compute_acc(y_q=y_q)

    def run_tables(self, table_s, s_idx, table_q, q_idx, cols, y_s, y_q, n_batches=1):
        """run_method for the task-batch loop (Evaluator_few_shot.evaluate_tasks): the support / query rows of task t are
        table_s[s_idx[t]] / table_q[q_idx[t]] with the columns permuted by cols[t] (Tasks_Generator_few_shot.get_task's
        `data[:, unique_labels]`), y_s / y_q the re-indexed labels.  The (T,S,K) support tensor - 16 MB per task at
        K = 1000 with 4 shots - is never built."""
        self._run_engine(None, None, y_s, n_batches=n_batches,
                         tables=dict(table_q=table_q, q_idx=q_idx, table_s=table_s, s_idx=s_idx, cols=cols))
        self.compute_acc(y_q=y_q)
