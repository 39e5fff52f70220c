"""Drop-in mirror of the reference's ``ops`` module surface, backed by HIP kernels.

Same names, argument order and error behaviour as /root/reference/ops.py (cited per
function); the arithmetic runs in libitcv_hip.so (hipvae.functional).  Differences that a
caller can observe are limited to:
  * ``total_correlation`` never materialises the [B,B,D] pairwise tensor (fused kernels; ``tc_components`` /
    ``tc_decomposition`` expose the fused estimator); the reference's building blocks that produce or take that
    tensor -- ``gaussian_log_density_torch``, ``gaussian_log_density``, ``minibatch_stratified_sampling``,
    ``minibatch_weighted_sampling``, ``on_off_diag`` -- exist under their own names in materialising form, so code
    that imports them (solvers/tc.py:5-11) keeps working;
  * the N(0,1) draws of ``reparameterize`` come from ``noise`` (device generator by default;
    ``set_noise_mode("host")`` reproduces the reference's CPU stream, ``noise_queue`` injects
    recorded draws for parity tests).
"""
import contextlib
import math

import numpy as np
import torch

from hipvae import abi
from hipvae import functional as HF

# ---------------------------------------------------------------------------- noise source
_noise = {"mode": "device", "queue": None}


def set_noise_mode(mode):
    """'device': torch.randn on the tensor's device (ops.py:184 on a GPU run);
    'host': CPU generator then copy (bit-identical to the reference's --device -1 stream)."""
    assert mode in ("device", "host")
    _noise["mode"] = mode


@contextlib.contextmanager
def noise_queue(draws):
    """Feed recorded N(0,1) tensors, in draw order, to every ``noise()`` call inside the block."""
    prev = _noise["queue"]
    _noise["queue"] = list(draws)
    try:
        yield
    finally:
        _noise["queue"] = prev


def noise(shape, device):
    q = _noise["queue"]
    if q is not None:
        if not q:
            raise RuntimeError("noise_queue exhausted")
        t = q.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.to(device=device, dtype=torch.float32)
    if _noise["mode"] == "host":
        return torch.randn(shape).to(device)
    return torch.randn(shape, device=device)


# ---------------------------------------------------------------------------- ops surface
def reparameterize(mu, logvar):
    """ops.py:166-185."""
    return HF.ReparamFn.apply(mu, logvar, noise(mu.shape, mu.device))


def kl_no_reduce(logvar, mu):
    """ops.py:161-163 (argument order: logvar, mu)."""
    return HF.KlRowsFn.apply(logvar, mu)


def kl_divergence(logvar, mu, reduce="sum", scale=1.0):
    """ops.py:136-158 (any ``reduce`` other than 'sum' / 'mean' returns the per-sample vector, as there), in ONE launch;
    ``scale`` (extension) multiplies the result inside that launch -- the solver hooks pass their beta."""
    return HF.KlLossFn.apply(logvar, mu, {"sum": 1, "mean": 2}.get(reduce, 0), float(scale))


def reconstruction_loss(x, recon_x, loss_type="mse", reduction="sum", scale=1.0):
    """ops.py:188-236; NotImplementedError for unknown loss_type / reduction, AssertionError on
    an empty batch, x is detached.  The reduction runs inside the kernels (two launches in all); ``scale`` (extension)
    multiplies the result there -- the solver hooks pass their beta."""
    batch_size = x.size(0)
    assert batch_size != 0
    if reduction not in ("sum", "mean", "none"):
        raise NotImplementedError
    if loss_type not in abi.LOSS_TYPES:
        raise NotImplementedError
    return HF.ReconLossFn.apply(x.detach().reshape(x.size(0), -1), recon_x.reshape(recon_x.size(0), -1),
                                abi.LOSS_TYPES[loss_type], HF._REDUCTION[reduction], float(scale))


def gaussian_log_density_torch(x, mu, logvar):
    """ops.py:15-21: -gaussian_nll_loss(x, mu, exp(logvar), eps=1e-4, full=True) clamped at -50, elementwise over the
    broadcast of the three operands; the variance floor passes the gradient straight through (as F.gaussian_nll_loss
    does), the -50 clamp blocks it."""
    return HF.GaussLogDensityFn.apply(x, mu, logvar, True)


def gaussian_log_density(x, mu, logvar):
    """ops.py:24-29: -0.5 * ((x-mu)^2 exp(-logvar) + logvar + log 2 pi) clamped at -50."""
    return HF.GaussLogDensityFn.apply(x, mu, logvar, False)


def minibatch_weighted_sampling(log_qz_prob, batch_size, dataset_size):
    """ops.py:92-101 on a materialised [B,B,D] tensor -> (logqz_prodmarginals [B], log_qz [B])."""
    assert log_qz_prob.size(0) == batch_size
    return HF.SamplingFn.apply(log_qz_prob, int(dataset_size), True)


def minibatch_stratified_sampling(log_qz_prob, batch_size, dataset_size):
    """ops.py:104-115 on a materialised [B,B,D] tensor -> (logqz_prodmarginals [B], log_qz [B])."""
    assert log_qz_prob.size(0) == batch_size
    return HF.SamplingFn.apply(log_qz_prob, int(dataset_size), False)


def on_off_diag(x):
    """ops.py:118-122 (never called by the reference): (torch.diagonal(x), x - torch.diag_embed(x)) for a 2-D x.
    No gradient is recorded."""
    return HF.on_off_diag(x)


def total_correlation(z, mu, logvar, dataset_size, reduce="mean", mu_all=None, row_offset=0):
    """ops.py:52-89.  ``mu_all``/``row_offset`` are the data-parallel extension: the means of the
    whole global batch (all-gathered) and this rank's first global row."""
    tc = HF.TcRowsFn.apply(z, mu if mu_all is None else mu_all, logvar, int(dataset_size), int(row_offset))
    return tc.mean() if reduce == "mean" else tc


def tc_kl_loss(z, mu, logvar, dataset_size, beta, reduce="mean", mu_all=None, row_offset=0):
    """(beta - 1) * total_correlation(z, mu, logvar, N, reduce) + kl_divergence(logvar, mu, reduce): the KL hook of the TC
    solvers (solvers/tc.py:69-89) fused into the estimator's launches (3 forward, 2 backward).  ``reduce`` as in the
    reference: 'mean', else per sample."""
    return HF.TcKlFn.apply(z, mu if mu_all is None else mu_all, logvar, int(dataset_size), int(row_offset), float(beta) - 1.0,
                           1.0, 2 if reduce == "mean" else 0)


def tc_components(z, mu, logvar, dataset_size, *, var_from_row=True, eps_density=True, weighted=False):
    """Fused ops.py:80-84 + :92-115: returns (logqz_prodmarginals [B], log_qz [B]) of the stratified
    (default) or weighted sampler, for either density flavour / variance orientation.  No grad."""
    flags = (abi.TC_VAR_FROM_ROW if var_from_row else 0) | (abi.TC_EPS_DENSITY if eps_density else 0) | (
        abi.TC_WEIGHTED if weighted else 0)
    with torch.no_grad():
        prodm, logqz, _ = HF.tc_components(z, mu, logvar, dataset_size, 0, flags)
    return prodm, logqz


def tc_decomposition(z, mu, logvar, dataset_size):
    """solvers/tc.py:104-121 per-sample (mi, tc, dwkl): the dead-code decomposition of the reference
    (un-eps'd density, variance indexed by component), offered as metrics."""
    with torch.no_grad():
        logq_cx, logpz = HF.diag_logdensity_rows(z, mu, logvar)
        prodm, logqz, _ = HF.tc_components(z, mu, logvar, dataset_size, 0, 0)
    return logq_cx - logqz, logqz - prodm, prodm - logpz


def log_importance_weight_matrix(batch_size, dataset_size):
    """ops.py:32-49 (host helper; the kernels evaluate these three values inline)."""
    n, m = dataset_size, batch_size - 1
    strat = (n - m) / (n * m)
    w = np.full((batch_size, batch_size), 1.0 / m, dtype=np.float32)
    w[:, 0] = 1.0 / n
    w[:, 1] = strat
    w[m - 1, 0] = strat
    return torch.from_numpy(np.log(w))


def entropy(x, base=None, axis=0, eps=1e-9):
    """ops.py:125-133 (numpy, evaluation only)."""
    if not isinstance(x, np.ndarray):
        raise TypeError("Input x has to be a numpy.ndarray object!")
    shifted = x + eps
    prob = shifted / shifted.sum(axis=axis, keepdims=True)
    h = -(prob * np.log(prob + eps)).sum(axis=axis)
    return h if base is None else h / math.log(base + eps)
