"""Few-shot ALPHA_TIM on probability features, drop-in for the reference's src/methods/few_shot/tim.py:192-322
(SURVEY.md F4).  Same constructor / run_task / logs contract; the `iter` Adam steps run in libtclip.so
(tclip_alpha_tim_run) with the gradient in closed form instead of autograd.  The reference's MKL matmuls and autograd
accumulation order leave no bit-level target, so this class is pinned to the reference within a float tolerance
(tests/test_alpha_tim.py).  TIM_GD (tim.py:91-189) is not reachable from the reference's evaluator and is not provided."""
import time

import torch

from src.methods._em_dirichlet_base import EMDirichletBase, FewShotMixin
from tclip_amd import engine


class BASE(FewShotMixin, EMDirichletBase):
    FEW_SHOT = True


class ALPHA_TIM(BASE):
    BANNER = "ALPHA_TIM"

    def __init__(self, model, device, log_file, args):
        if not hasattr(args, "iter_mm"):
            args.iter_mm = 0          # alpha_tim.yaml has no iter_mm
        if not hasattr(args, "k_eff"):
            args.k_eff = 5            # only feeds the unused EM-Dirichlet lambd of the shared base
        super().__init__(model=model, device=device, log_file=log_file, args=args)
        self.loss_weights = list(args.loss_weights)      # tim.py:29 (.copy())
        self.temp = args.temp
        self.lr = float(args.lr_alpha_tim)               # tim.py:196
        self.entropies = list(args.entropies)
        self.alpha_value = args.alpha_value

    def run_method(self, support, query, y_s, y_q, n_batches=1):
        if query.shape[2] != self.args.num_classes_test:
            raise NotImplementedError("ALPHA_TIM here takes probability features (feature dimension = n_class)")
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("ALPHA_TIM on MI355X needs device='cuda': there is no CPU fallback in this package")
        self.logger.info(" ==> Executing ALPHA_TIM with ALPHA = {} and T = {}".format(self.alpha_value, self.args.T))
        n_task = query.shape[0]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        self.weights, self.logits_q, self.preds, crit = engine.run_alpha_tim(
            query, support, y_s, iters=self.iter, temp=self.temp, lr=self.lr, alpha_value=self.alpha_value,
            loss_weights=self.loss_weights, entropies=self.entropies, n_batches=n_batches)
        torch.cuda.synchronize(dev)
        total = time.time() - t0
        for i in range(self.iter):
            # cumulative wall time per iteration over n_task (tim.py:317-319)
            self.timestamps.append(total * (i + 1) / max(self.iter, 1) / n_task)
        crit = crit.cpu().numpy()                        # (n_batches, iter): mean_{task,class} ||w_old - w||
        self.criterions_per_batch = crit
        self.criterions = list(crit.mean(0)) if n_batches > 1 else list(crit[0])
        self.compute_acc(y_q=y_q)                        # argmax of the last iteration's query logits (tim.py:321)
