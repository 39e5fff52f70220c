"""Oracle restatement of the reference's ops.py (test infrastructure, see oracle/__init__).

All functions are dtype-generic torch code on CPU tensors, written as explicit formulas
so they can be evaluated in fp64 as a tight check.  Reference lines are cited per function
as /root/reference/<file>:<line>.
"""
import math

import torch

LOG_2PI = math.log(2.0 * math.pi)
VAR_EPS = 1e-4      # ops.py:18  (eps passed to F.gaussian_nll_loss)
LOGP_FLOOR = -50.0  # ops.py:21,29


def reparameterize(mu, logvar, eps):
    """ops.py:166-185 with the N(0,1) draw made an explicit input."""
    return mu + eps * torch.exp(0.5 * logvar)


def kl_rows(logvar, mu):
    """ops.py:161-163  KL(q(z|x) || N(0,I)) per sample -> [B]."""
    return -0.5 * (1.0 + logvar - logvar.exp() - mu * mu).sum(1)


def kl(logvar, mu, reduce="sum"):
    """ops.py:136-158 (argument order logvar, mu)."""
    r = kl_rows(logvar, mu)
    if reduce == "sum":
        return r.sum()
    if reduce == "mean":
        return r.mean()
    return r


def log_density_clamped_var(x, m, logvar):
    """ops.py:15-21.  -gaussian_nll_loss(full=True, eps=1e-4) then clamp(min=-50).

    torch's gaussian_nll_loss clamps the variance on a clone under no_grad, so the VALUE
    uses max(var, eps) while the GRADIENT w.r.t. var is that of the unclamped expression
    evaluated at the clamped value (straight-through)."""
    var = torch.exp(logvar)
    vhat = var + (var.clamp(min=VAR_EPS) - var).detach()
    lp = -(0.5 * (torch.log(vhat) + (x - m) ** 2 / vhat) + 0.5 * LOG_2PI)
    return lp.clamp(min=LOGP_FLOOR)


def log_density_plain(x, m, logvar):
    """ops.py:24-29 (dead-code density used only by solvers/tc.py:91-144)."""
    d = x - m
    lp = -0.5 * (d * d * torch.exp(-logvar) + logvar + LOG_2PI)
    return lp.clamp(min=LOGP_FLOOR)


def log_importance_weights(batch, dataset_size, dtype=torch.float32):
    """ops.py:32-49.  NB the flat stride M+1 == batch addresses COLUMNS 0 and 1 of every
    row (not the diagonal); then element [M-1, 0] is overwritten.  Built in fp32 like the
    reference (torch.Tensor(...).fill_), then log."""
    n, m = dataset_size, batch - 1
    strat = (n - m) / (n * m)
    w = torch.full((batch, batch), 1.0 / m, dtype=torch.float32)
    w[:, 0] = 1.0 / n
    if batch > 1:
        w[:, 1] = strat
    w[m - 1, 0] = strat
    return w.log().to(dtype)


def stratified(lp, dataset_size):
    """ops.py:104-115.  lp [j,i,l] -> (sum_l logsumexp_i(logW+lp), logsumexp_i(logW+sum_l lp))."""
    b = lp.shape[0]
    lw = log_importance_weights(b, dataset_size, lp.dtype)
    prodm = torch.logsumexp(lw.view(b, b, 1) + lp, dim=1).sum(1)
    logqz = torch.logsumexp(lw + lp.sum(2), dim=1)
    return prodm, logqz


def weighted(lp, dataset_size):
    """ops.py:92-101 (minibatch-weighted sampling; never called by the live loss)."""
    b = lp.shape[0]
    c = math.log(b * dataset_size)
    prodm = (torch.logsumexp(lp, dim=1) - c).sum(1)
    logqz = torch.logsumexp(lp.sum(2), dim=1) - c
    return prodm, logqz


def pairwise_live(z, mu, logvar):
    """ops.py:80-82.  [j,i,l] = log q(z_j | mu_i, var_J) with the variance taken from the
    SAMPLE row j (logvar.unsqueeze(1)) -- the reference's transposed-variance quirk."""
    return log_density_clamped_var(z.unsqueeze(1), mu.unsqueeze(0), logvar.unsqueeze(1))


def total_correlation(z, mu, logvar, dataset_size, reduce="mean"):
    """ops.py:52-89."""
    prodm, logqz = stratified(pairwise_live(z, mu, logvar), dataset_size)
    tc = logqz - prodm
    return tc.mean() if reduce == "mean" else tc


def tc_kl(z, mu, logvar, dataset_size, beta, reduce="mean"):
    """solvers/tc.py:69-89  (beta-1)*TC + analytic KL."""
    return (beta - 1.0) * total_correlation(z, mu, logvar, dataset_size, reduce) + kl(logvar, mu, reduce)


def decomposition(z, mu, logvar, dataset_size):
    """solvers/tc.py:104-121 per-sample (mi, tc, dwkl) with the un-eps'd density and the
    variance indexed by component i (logvar.unsqueeze(0))."""
    logq_cx = log_density_plain(z, mu, logvar).sum(1)
    zeros = torch.zeros_like(z)
    logpz = log_density_plain(z, zeros, zeros).sum(1)
    lp = log_density_plain(z.unsqueeze(1), mu.unsqueeze(0), logvar.unsqueeze(0))
    prodm, logqz = stratified(lp, dataset_size)
    return logq_cx - logqz, logqz - prodm, prodm - logpz


def reconstruction_rows(x, recon, loss_type):
    """ops.py:219-230  per-sample summed error -> [B]; x is detached (ops.py:220)."""
    r = recon.reshape(recon.shape[0], -1)
    t = x.reshape(x.shape[0], -1).detach()
    if loss_type == "mse":
        e = (r - t) ** 2
    elif loss_type == "l1":
        e = (r - t).abs()
    elif loss_type == "bce":
        # F.binary_cross_entropy clamps each log term at -100
        e = -(t * torch.log(r).clamp(min=-100.0) + (1.0 - t) * torch.log(1.0 - r).clamp(min=-100.0))
    else:
        raise NotImplementedError(loss_type)
    return e.sum(1)


def reconstruction_loss(x, recon, loss_type="mse", reduction="sum"):
    """ops.py:188-236."""
    assert x.shape[0] != 0
    if reduction not in ("sum", "mean", "none"):
        raise NotImplementedError(reduction)
    rows = reconstruction_rows(x, recon, loss_type)
    if reduction == "sum":
        return rows.sum()
    if reduction == "mean":
        return rows.mean()
    return rows
