"""Zero-shot EM_GAUSSIAN_COV on probability features, drop-in for the reference's
src/methods/zero_shot/em_gaussian_cov.py: EM_GAUSSIAN with a diagonal inverse
covariance per cluster and no temperature.  Same constructor / run_task / logs contract; the loop
runs in libtclip.so (tclip_em_gaussian_cov_run).  Visual (non-simplex) features need CLIP text
prompts for the initial assignment (reference :219-229) and are out of scope."""
import time

import torch

from src.methods._em_dirichlet_base import EMDirichletBase, ZeroShotMixin
from tclip_amd import engine


class BASE(ZeroShotMixin, EMDirichletBase):
    pass


class EM_GAUSSIAN_COV(BASE):
    BANNER = "EM_GAUSSIAN_COV"

    def __init__(self, model, device, log_file, args):
        if not hasattr(args, "iter_mm"):
            args.iter_mm = 0          # em_gaussian_cov.yaml has no iter_mm
        super().__init__(model=model, device=device, log_file=log_file, args=args)     # lambd = int(K/5) * n_query (:20)

    def run_method(self, query, y_q, n_batches=1):
        if not self.args.use_softmax_feature:
            raise NotImplementedError("EM_GAUSSIAN_COV on visual features needs CLIP text prompts (out of scope)")
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("EM_GAUSSIAN_COV on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing EM_GAUSSIAN_COV with T = {}".format(self.args.T))
        n_task = query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.u, self.v, self.w, self.s, self.preds = engine.run_em_gaussian_cov(query, iters=self.iter, lambd=self.lambd)
        torch.cuda.synchronize(dev)
        total = time.time() - t0
        for i in range(self.iter):
            # the reference restarts its clock every iteration (em_gaussian_cov.py:234-254)
            self.timestamps.append(total / max(self.iter, 1) / n_task)
        self.criterions = [0.0] * self.iter       # the reference compares u with a copy of itself
        self.compute_acc_clustering(query, y_q)
