"""IntroTCSovler (sic): the Soft-Intro step with the beta-TC KL hook
(/root/reference/solvers/intro_tc.py:7-17) -- the configuration the headline metric is quoted on."""
from typing import Optional

from torch import Tensor

from solvers.intro import IntroSolver
from solvers.tc import TCSovler


class IntroTCSovler(IntroSolver):
    def compute_kl_loss(self, z: Optional[Tensor], mu: Tensor, logvar: Tensor, reduce: str = "mean",
                        beta: float = None, write: bool = False) -> Tensor:
        return TCSovler.compute_kl_loss(self, z, mu, logvar, reduce, beta, write)

    kl_decomposition = TCSovler.kl_decomposition
