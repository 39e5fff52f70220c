"""TCSovler (sic): VAESolver whose KL hook is the beta-TC-VAE term (beta-1)*TC + KL
(/root/reference/solvers/tc.py:22-89), plus the (mi, tc, dwkl) decomposition of the
reference's dead ``_compute_kl_loss_full`` (tc.py:91-144) offered as metrics."""
from typing import Optional

from torch import Tensor

from hipvae import ddp
from ops import kl_divergence, tc_decomposition, tc_kl_loss, total_correlation
from solvers.vae import VAESolver
from utils import SingletonWriter


class TCSovler(VAESolver):
    def compute_kl_loss(self, z: Optional[Tensor], mu: Tensor, logvar: Tensor, reduce: str = "mean",
                        beta: float = None, write: bool = False) -> Tensor:
        return TCSovler._compute_kl_loss_simple(self, z, mu, logvar, reduce, beta, write)

    def _compute_kl_loss_simple(self, z, mu, logvar, reduce="mean", beta=None, write=False) -> Tensor:
        """tc.py:69-89.  In a data-parallel run the estimator sees the whole global batch: the means
        are all-gathered (the variance is the sample row's own, ops.py:81) and the importance weights
        use global batch size and row indices."""
        if beta is None:
            beta = self.beta_kl
        dataset_size = len(self.dataset)
        if not (write and self.writer):       # the whole hook inside the estimator's own launches
            return tc_kl_loss(z, mu, logvar, dataset_size, beta, reduce, mu_all=ddp.all_gather_rows(mu),
                              row_offset=ddp.row_offset(mu.shape[0]))
        kl_loss = kl_divergence(logvar, mu, reduce=reduce)
        tc = total_correlation(z, mu, logvar, dataset_size, reduce=reduce, mu_all=ddp.all_gather_rows(mu),
                               row_offset=ddp.row_offset(mu.shape[0]))
        self.write_scalar(SingletonWriter().cur_iter, "kl_loss_unscaled", kl_loss)
        return (beta - 1.0) * tc + kl_loss

    def kl_decomposition(self, z, mu, logvar):
        """Per-sample (mi, tc, dwkl) of tc.py:104-121 as metrics (no gradient, single-rank batch)."""
        return tc_decomposition(z, mu, logvar, len(self.dataset))
