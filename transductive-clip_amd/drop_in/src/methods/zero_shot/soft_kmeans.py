"""Zero-shot SOFT_KMEANS on probability features, drop-in for the reference's
src/methods/zero_shot/soft_kmeans.py (BASELINE config 3's second method).
Same constructor / run_task / logs contract as the EM-Dirichlet classes; the loop runs in
libtclip.so (tclip_soft_kmeans_run).  Visual (non-simplex) features need CLIP text prompts for the
initial assignment (reference :187-197) and are out of scope."""
import time

import torch

from src.methods._em_dirichlet_base import EMDirichletBase, ZeroShotMixin
from tclip_amd import engine


class BASE(ZeroShotMixin, EMDirichletBase):
    pass


class SOFT_KMEANS(BASE):
    BANNER = "SOFT K-MEANS"

    def __init__(self, model, device, log_file, args):
        if not hasattr(args, "iter_mm"):
            args.iter_mm = 0          # soft_kmeans.yaml has no iter_mm
        super().__init__(model=model, device=device, log_file=log_file, args=args)

    def run_method(self, query, y_q, n_batches=1):
        if not self.args.use_softmax_feature:
            raise NotImplementedError("SOFT_KMEANS on visual features needs CLIP text prompts (out of scope)")
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("SOFT_KMEANS on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing SOFT K-MEANS with T = {}".format(self.args.T))
        n_task = query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.u, self.w, self.preds = engine.run_soft_kmeans(query, iters=self.iter, temperature=self.args.T)
        torch.cuda.synchronize(dev)
        total = time.time() - t0
        for i in range(self.iter):
            # the reference restarts its clock every iteration (soft_kmeans.py:203-216)
            self.timestamps.append(total / max(self.iter, 1) / n_task)
        self.criterions = [0.0] * self.iter       # the reference compares u with a copy of itself
        self.compute_acc_clustering(query, y_q)
