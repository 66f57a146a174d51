"""Few-shot BD-CSPN (prototype rectification) on probability features, drop-in for the reference's
src/methods/few_shot/bdcspn.py.  Same constructor / run_task / logs contract
(args.norm_type, args.temp, args.n_class; one timestamp, one zero criterion, plain accuracy); the
whole pass runs in libtclip.so (tclip_bdcspn_run), all tasks of the batch at once instead of the
reference's per-task Python loop (:124-141)."""
import time

import numpy as np
import torch

from src.utils import Logger
from tclip_amd import engine


class BDCSPN(object):
    def __init__(self, model, device, log_file, args):
        self.device = device
        self.norm_type = args.norm_type
        self.temp = args.temp
        self.model = model
        self.log_file = log_file
        self.n_class = args.n_class
        self.logger = Logger(__name__, self.log_file)
        self.init_info_lists()

    def __del__(self):
        try:
            self.logger.del_logger()
        except Exception:
            pass

    def init_info_lists(self):
        self.timestamps = []
        self.criterions = []
        self.test_acc = []

    def record_convergence(self, new_time, criterions):
        self.criterions.append(criterions)
        self.timestamps.append(new_time)

    def compute_acc(self, y_q, preds_q):
        # on the host: the mean of 75 zeros and ones is rounded as the reference's CPU op rounds it
        accuracy = (preds_q.long().cpu() == y_q.cpu()).float().mean(1, keepdim=True)
        self.test_acc.append(accuracy)

    def get_logs(self):
        self.criterions = torch.stack(self.criterions, dim=0).cpu().numpy()
        self.test_acc = torch.cat(self.test_acc, dim=1).cpu().numpy()
        return {'timestamps': np.array(self.timestamps).mean(), 'criterions': self.criterions,
                'acc': self.test_acc}

    def run_task(self, task_dic, shot=None):
        y_s, y_q = task_dic['y_s'], task_dic['y_q']
        support, query = task_dic['x_s'], task_dic['x_q']
        support = support.to(self.device).float()
        query = query.to(self.device).float()
        y_s = y_s.long().squeeze(2).to(self.device)
        y_q = y_q.long().squeeze(2).to(self.device)
        # the reference normalises here (:165-166) and hands the result to run_method; the engine does both
        self.run_batch(support=support, query=query, y_s=y_s, y_q=y_q)
        return self.get_logs()

    def run_method(self, support, query, y_s, y_q, shot=None, n_batches=1):
        """Reference semantics (:172-200): `support` and `query` are already normalised."""
        self.run_batch(support, query, y_s, y_q, norm_type="UN")

    def run_batch(self, support, query, y_s, y_q, n_batches=1, norm_type=None):
        """Normalisation + BD-CSPN for all tasks at once (what run_task does for one batch)."""
        norm_type = self.norm_type if norm_type is None else norm_type
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("BDCSPN on MI355X needs device='cuda': there is no CPU fallback in this package")
        if query.shape[2] != self.n_class:
            raise NotImplementedError("BDCSPN here takes probability features (feature dimension = n_class)")
        self.logger.info(" ==> Executing BD-CSPN")
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.prototypes, self.u, self.preds = engine.run_bdcspn(query, support, y_s, temp=self.temp, norm_type=norm_type)
        torch.cuda.synchronize(dev)
        self.record_convergence(new_time=time.time() - t0, criterions=torch.zeros(1))
        self.compute_acc(y_q, self.preds)
