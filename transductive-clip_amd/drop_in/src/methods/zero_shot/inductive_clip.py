"""Inductive zero-shot CLIP on probability features, drop-in for the reference's
src/methods/zero_shot/inductive_clip.py (the baseline every transductive method is compared with):
no adaptation, u = the query features, prediction = their arg-max, plain accuracy (no cluster
matching).  The arg-max runs in libtclip.so (tclip_argmax_rows).  Visual features need CLIP text
prompts (reference :115-124) and are out of scope."""
import numpy as np
import torch

from src.utils import Logger
from tclip_amd import engine


class BASE(object):
    def __init__(self, model, device, log_file, args):
        self.device = device
        self.model = model
        self.log_file = log_file
        self.logger = Logger(__name__, self.log_file)
        self.init_info_lists()
        self.args = args

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

    def compute_acc(self, y_q):
        self.preds = engine.argmax_rows(self.u)
        # on the host: the mean of 75 zeros and ones is rounded as the reference's CPU op rounds it
        accuracy = (self.preds.long().cpu() == y_q.cpu()).float().mean(1, keepdim=True)
        self.test_acc.append(accuracy)

    def get_logs(self):
        self.criterions = torch.stack(self.criterions, dim=0).cpu().numpy()
        self.test_acc = torch.cat(self.test_acc, dim=1).cpu().numpy()
        return {'timestamps': np.array(self.timestamps).mean(), 'criterions': self.criterions,
                'acc': self.test_acc}

    def run_task(self, task_dic):
        y_q, query = task_dic['y_q'], task_dic['x_q']
        query = query.to(self.device).float()
        y_q = y_q.long().squeeze(2).to(self.device)
        self.run_method(query=query, y_q=y_q)
        return self.get_logs()


class CLIP(BASE):
    def run_method(self, query, y_q, n_batches=1):
        if not self.args.use_softmax_feature:
            raise NotImplementedError("CLIP on visual features needs CLIP text prompts (out of scope)")
        if torch.device(self.device).type != "cuda":
            raise RuntimeError("CLIP on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing CLIP")
        self.u = query
        self.record_convergence(new_time=0, criterions=torch.zeros(()))      # ||u - copy of u|| (:126-128)
        self.compute_acc(y_q)
