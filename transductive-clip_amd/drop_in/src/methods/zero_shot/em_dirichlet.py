"""Zero-shot EM-Dirichlet, drop-in for the reference's src/methods/zero_shot/em_dirichlet.py
(class names, constructor keywords, run_task and its logs dict are the reference's)."""
from src.methods._em_dirichlet_base import EMDirichletBase, ZeroShotMixin


class BASE(ZeroShotMixin, EMDirichletBase):
    pass


class EM_DIRICHLET(BASE):
    HARD = False
    BANNER = "EM-DIRICHLET"

    def __init__(self, model, device, log_file, args):
        super().__init__(model=model, device=device, log_file=log_file, args=args)
