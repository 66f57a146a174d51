"""Few-shot EM-Dirichlet, drop-in for the reference's src/methods/few_shot/em_dirichlet.py."""
from src.methods._em_dirichlet_base import EMDirichletBase, FewShotMixin


class BASE(FewShotMixin, EMDirichletBase):
    FEW_SHOT = True


class EM_DIRICHLET(BASE):
    HARD = False
    BANNER = "EM-DIRICHLET"

    def __init__(self, model, device, log_file, args):
        super().__init__(model=model, device=device, log_file=log_file, args=args)
