"""Zero-shot Hard EM-Dirichlet, drop-in for src/methods/zero_shot/hard_em_dirichlet.py
(one-hot responsibilities after every E-step, reference :255-258)."""
from src.methods._em_dirichlet_base import EMDirichletBase, ZeroShotMixin


class BASE(ZeroShotMixin, EMDirichletBase):
    pass


class HARD_EM_DIRICHLET(BASE):
    HARD = True
    BANNER = "HARD EM-DIRICHLET"

    def __init__(self, model, device, log_file, args):
        super().__init__(model=model, device=device, log_file=log_file, args=args)
