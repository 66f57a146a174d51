"""Zero-shot HARD_KMEANS on probability features, drop-in for the reference's
src/methods/zero_shot/hard_kmeans.py.  Same constructor / run_task / logs contract
as the EM-Dirichlet classes; the loop runs in libtclip.so (tclip_hard_kmeans_run).  Visual
(non-simplex) features need CLIP text prompts for the initial assignment (reference :173-183) and
are out of scope."""
import time

import torch

from src.methods._em_dirichlet_base import EMDirichletBase, ZeroShotMixin
from tclip_amd import engine


class BASE(ZeroShotMixin, EMDirichletBase):
    pass


class HARD_KMEANS(BASE):
    BANNER = "HARD_KMEANS"

    def __init__(self, model, device, log_file, args):
        if not hasattr(args, "iter_mm"):
            args.iter_mm = 0          # hard_kmeans.yaml has no iter_mm
        super().__init__(model=model, device=device, log_file=log_file, args=args)

    def run_method(self, query, y_q, n_batches=1):
        if not self.args.use_softmax_feature:
            raise NotImplementedError("HARD_KMEANS on visual features needs CLIP text prompts (out of scope)")
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("HARD_KMEANS on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing HARD_KMEANS with T = {}".format(self.args.T))
        n_task = query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.u, self.w, self.preds, crit = engine.run_hard_kmeans(query, iters=self.iter, n_batches=n_batches)
        crit = crit.cpu()
        total = time.time() - t0
        self.criterions = []
        for i in range(self.iter):
            # the reference records every iteration twice (hard_kmeans.py:198-204)
            for rep in range(2):
                self.timestamps.append(total / max(self.iter, 1) / (1 if rep == 0 else n_task))
                self.criterions.append(crit[0, i])
        self.compute_acc_clustering(query, y_q)
