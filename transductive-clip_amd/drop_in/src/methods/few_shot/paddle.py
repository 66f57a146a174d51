"""Few-shot PADDLE on probability features, drop-in for the reference's
src/methods/few_shot/paddle.py.  Same constructor / run_task / logs contract; the
loop runs in libtclip.so (tclip_paddle_run).  `args.lambd` is the method's own float (paddle.yaml),
not the class-count formula of EM-Dirichlet.  Visual (non-simplex) features need CLIP text prompts
for the initial assignment (reference :186-196) and are out of scope."""
import time

import torch

from src.methods._em_dirichlet_base import EMDirichletBase, FewShotMixin
from tclip_amd import engine


class BASE(FewShotMixin, EMDirichletBase):
    FEW_SHOT = True


class PADDLE(BASE):
    BANNER = "PADDLE"

    def __init__(self, model, device, log_file, args):
        if not hasattr(args, "iter_mm"):
            args.iter_mm = 0          # paddle.yaml has no iter_mm
        super().__init__(model=model, device=device, log_file=log_file, args=args)
        self.lambd = args.lambd       # paddle.py:26

    def run_method(self, support, query, y_s, y_q, n_batches=1):
        if not self.args.use_softmax_feature:
            raise NotImplementedError("PADDLE on visual features needs CLIP text prompts (out of scope)")
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("PADDLE on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing PADDLE with LAMBDA = {} and T = {}".format(self.lambd, self.args.T))
        n_task = query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.u, self.v, self.w, self.preds = engine.run_paddle(query, support, y_s, iters=self.iter, lambd=self.lambd)
        torch.cuda.synchronize(dev)
        total = time.time() - t0
        for i in range(self.iter):
            # cumulative wall time per iteration over n_task (paddle.py:214-216)
            self.timestamps.append(total * (i + 1) / max(self.iter, 1) / n_task)
        self.criterions = [0.0] * self.iter       # the reference compares u with a copy of itself (:211-212)
        self.compute_acc(y_q=y_q)
