"""Few-shot Hard EM-Dirichlet, drop-in for src/methods/few_shot/hard_em_dirichlet.py.
Its logged criterions are identically 0, as in the reference (:233-244 evaluates the criterion
after alpha_old has been refreshed)."""
from src.methods._em_dirichlet_base import EMDirichletBase, FewShotMixin


class BASE(FewShotMixin, EMDirichletBase):
    FEW_SHOT = True


class HARD_EM_DIRICHLET(BASE):
    HARD = True
    BANNER = "HARD EM-DIRICHLET"

    def __init__(self, model, device, log_file, args):
        super().__init__(model=model, device=device, log_file=log_file, args=args)
