"""Zero-shot EM_GAUSSIAN on probability features, drop-in for the reference's
src/methods/zero_shot/em_gaussian.py: SOFT_KMEANS plus the class-proportion term of
EM-Dirichlet.  Same constructor / run_task / logs contract; the loop runs in libtclip.so
(tclip_em_gaussian_run).  Visual (non-simplex) features need CLIP text prompts for the initial
assignment (reference :188-198) and are out of scope."""
import time

import torch

from src.methods._em_dirichlet_base import EMDirichletBase, ZeroShotMixin
from tclip_amd import engine


class BASE(ZeroShotMixin, EMDirichletBase):
    pass


class EM_GAUSSIAN(BASE):
    BANNER = "EM_GAUSSIAN"

    def __init__(self, model, device, log_file, args):
        if not hasattr(args, "iter_mm"):
            args.iter_mm = 0          # em_gaussian.yaml has no iter_mm
        super().__init__(model=model, device=device, log_file=log_file, args=args)     # lambd = int(K/5) * n_query (:20)

    def run_method(self, query, y_q, n_batches=1):
        if not self.args.use_softmax_feature:
            raise NotImplementedError("EM_GAUSSIAN on visual features needs CLIP text prompts (out of scope)")
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("EM_GAUSSIAN on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing EM_GAUSSIAN with T = {}".format(self.args.T))
        n_task = query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.u, self.v, self.w, self.preds = engine.run_em_gaussian(query, iters=self.iter, temperature=self.args.T,
                                                                    lambd=self.lambd)
        torch.cuda.synchronize(dev)
        total = time.time() - t0
        for i in range(self.iter):
            # the reference restarts its clock every iteration (em_gaussian.py:204-224)
            self.timestamps.append(total / max(self.iter, 1) / n_task)
        self.criterions = [0.0] * self.iter       # the reference compares u with a copy of itself
        self.compute_acc_clustering(query, y_q)
