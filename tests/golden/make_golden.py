#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the *unmodified* reference (meffmadd/intro-tc-vae) from /root/reference with
in-process stub modules for packages that are absent in this image (black, torchvision,
tensorboard, xgboost), drives its ops / models / solvers on seeded inputs and writes
the inputs and the reference's outputs as small ``.npz`` fixtures next to this file.

Nothing from the reference is copied: the fixtures hold tensors only.  The fixtures are
what pins ``oracle/`` (tests/test_oracle_golden.py) and, through it, the HIP path.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Reference entry points exercised (file:line in /root/reference):
  ops.py:15-21,24-29,32-49,52-89,92-115,136-163,166-185,188-236
  models.py:196-355          solvers/vae.py:89-136   solvers/intro.py:56-196
  solvers/tc.py:58-144       solvers/intro_tc.py:7-17
"""
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    black = types.ModuleType("black")
    black.out = None
    sys.modules["black"] = black
    for name in (
        "torchvision", "torchvision.utils", "torchvision.transforms",
        "torchvision.transforms.functional", "torchvision.io", "torchvision.datasets",
        "torch.utils.tensorboard", "xgboost", "umap",
    ):
        sys.modules[name] = MagicMock()


_install_stubs()
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ops as R  # noqa: E402  (reference ops.py)
import models as RM  # noqa: E402
from solvers import VAESolver, IntroSolver  # noqa: E402
from solvers.tc import TCSovler  # noqa: E402
from solvers.intro_tc import IntroTCSovler  # noqa: E402
from utils import SingletonWriter  # noqa: E402

torch.set_num_threads(4)
F32 = np.float32


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- ops
def latent_inputs(B, D, seed):
    """logvar ~ N(-3, 3^2) so that the 1e-4 variance clamp (logvar < -9.21) and the
    -50 log-density clamp both fire on a visible fraction of the [B,B,D] elements."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(B, D, generator=g)
    logvar = -3.0 + 3.0 * torch.randn(B, D, generator=g)
    eps = torch.randn(B, D, generator=g)
    z = mu + eps * torch.exp(0.5 * logvar)
    return z, mu, logvar, eps


def latent_cases(cases, out):
    for tag, (B, D, N) in cases.items():
        z, mu, logvar, eps = latent_inputs(B, D, seed=100 + B)
        out[f"{tag}_BDN"] = np.array([B, D, N], dtype=np.int64)
        out[f"{tag}_z"], out[f"{tag}_mu"], out[f"{tag}_logvar"] = npy(z), npy(mu), npy(logvar)
        out[f"{tag}_eps"] = npy(eps)
        out[f"{tag}_kl_none"] = npy(R.kl_no_reduce(logvar, mu))
        out[f"{tag}_kl_sum"] = npy(R.kl_divergence(logvar, mu, "sum"))
        out[f"{tag}_kl_mean"] = npy(R.kl_divergence(logvar, mu, "mean"))
        out[f"{tag}_logiw"] = npy(R.log_importance_weight_matrix(B, N))
        # live estimator (ops.py:80-84): torch density, variance indexed by sample row j
        lp = R.gaussian_log_density_torch(z.unsqueeze(1), mu.unsqueeze(0), logvar.unsqueeze(1))
        out[f"{tag}_frac_clamp50"] = np.array([(lp <= -50).float().mean().item()], dtype=F32)
        out[f"{tag}_frac_varclamp"] = np.array([(logvar.exp() < 1e-4).float().mean().item()], dtype=F32)
        pm, lq = R.minibatch_stratified_sampling(lp, B, N)
        out[f"{tag}_mss_prodm"], out[f"{tag}_mss_logqz"] = npy(pm), npy(lq)
        pm, lq = R.minibatch_weighted_sampling(lp, B, N)
        out[f"{tag}_mws_prodm"], out[f"{tag}_mws_logqz"] = npy(pm), npy(lq)
        out[f"{tag}_tc_none"] = npy(R.total_correlation(z, mu, logvar, N, "none"))
        out[f"{tag}_tc_mean"] = npy(R.total_correlation(z, mu, logvar, N, "mean"))
        # gradients of tc.mean() and of ((beta-1) tc + kl).mean(), beta = 512 and 0.5
        for beta in (512.0, 0.5):
            zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, logvar))
            loss = ((beta - 1.0) * R.total_correlation(zz, mm, ll, N, "mean")
                    + R.kl_divergence(ll, mm, "mean"))
            loss.backward()
            bt = str(beta).replace(".", "p")
            out[f"{tag}_tckl_b{bt}"] = npy(loss)
            out[f"{tag}_tckl_b{bt}_dz"] = npy(zz.grad)
            out[f"{tag}_tckl_b{bt}_dmu"] = npy(mm.grad)
            out[f"{tag}_tckl_b{bt}_dlogvar"] = npy(ll.grad)
        # per-sample ("none") weighting: gradient of sum_j w_j * tc_j
        zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, logvar))
        w = torch.linspace(-1.0, 2.0, B)
        (w * R.total_correlation(zz, mm, ll, N, "none")).sum().backward()
        out[f"{tag}_tcw_w"] = npy(w)
        out[f"{tag}_tcw_dz"], out[f"{tag}_tcw_dmu"], out[f"{tag}_tcw_dlogvar"] = (
            npy(zz.grad), npy(mm.grad), npy(ll.grad))
        # dead-code full decomposition (solvers/tc.py:91-144): un-eps'd density,
        # variance indexed by component i (logvar.unsqueeze(0))
        logq_cx = R.gaussian_log_density(z, mu, logvar).sum(1)
        zeros = torch.zeros_like(z)
        logpz = R.gaussian_log_density(z, zeros, zeros).sum(1)
        lp2 = R.gaussian_log_density(z.unsqueeze(1), mu.unsqueeze(0), logvar.unsqueeze(0))
        pm2, lq2 = R.minibatch_stratified_sampling(lp2, B, N)
        out[f"{tag}_full_logq_cx"], out[f"{tag}_full_logpz"] = npy(logq_cx), npy(logpz)
        out[f"{tag}_full_prodm"], out[f"{tag}_full_logqz"] = npy(pm2), npy(lq2)
        out[f"{tag}_full_mi"] = npy(logq_cx - lq2)
        out[f"{tag}_full_tc"] = npy(lq2 - pm2)
        out[f"{tag}_full_dwkl"] = npy(pm2 - logpz)
        # reparameterize with the recorded eps (ops.py:183-185)
        out[f"{tag}_reparam"] = npy(mu + eps * torch.exp(0.5 * logvar))


def gen_ops_c4():
    """BASELINE configs[3] (c4): global batch 512, z_dim 128, N = 10000 -- the size SURVEY.md section 8(c) names.
    The reference materialises the [512,512,128] pairwise tensor (134 MB fp32) and its autograd copies: ~2 GB."""
    out = {}
    latent_cases({"d": (512, 128, 10000)}, out)
    save("ops_c4.npz", **out)


def gen_ops():
    out = {}
    latent_cases({"a": (16, 10, 1000), "b": (64, 128, 10000), "c": (256, 64, 10000)}, out)
    # reconstruction losses (ops.py:188-236) incl. the KATs of tests/test_ops.py:10-43
    g = torch.Generator().manual_seed(7)
    x = torch.rand(6, 3, 8, 8, generator=g)
    xr = torch.rand(6, 3, 8, 8, generator=g).clamp(1e-3, 1 - 1e-3)
    out["rec_x"], out["rec_xr"] = npy(x), npy(xr)
    for lt in ("mse", "l1", "bce"):
        for red in ("sum", "mean", "none"):
            out[f"rec_{lt}_{red}"] = npy(R.reconstruction_loss(x, xr, lt, red))
        xr_g = xr.clone().requires_grad_(True)
        w = torch.linspace(0.5, 1.5, 6)
        (w * R.reconstruction_loss(x, xr_g, lt, "none")).sum().backward()
        out[f"rec_{lt}_w"] = npy(w)
        out[f"rec_{lt}_dxr"] = npy(xr_g.grad)
    x0 = torch.tensor([0.0, 0.0, 0.0])
    x1 = torch.tensor([1.0, 2.0, 4.0])
    out["kat_mse"] = np.array([R.reconstruction_loss(x0, x1, "mse", r).sum().item() for r in ("sum", "mean")], dtype=F32)
    out["kat_mse_none"] = npy(R.reconstruction_loss(x0, x1, "mse", "none"))
    out["kat_l1"] = np.array([R.reconstruction_loss(x0, x1, "l1", r).item() for r in ("sum", "mean")], dtype=F32)
    save("ops.npz", **out)


# ------------------------------------------------------------------------- models
TINY = dict(cdim=3, zdim=10, channels=(8, 16, 32), image_size=32)


def build_model(arch, seed=0, **kw):
    torch.manual_seed(seed)
    cfg = dict(TINY)
    cfg.update(kw)
    return RM.SoftIntroVAE(arch=arch, **cfg)


def state_arrays(model, prefix):
    return {prefix + k.replace(".", "/"): npy(v) for k, v in model.state_dict().items()}


def gen_models():
    for arch in ("conv", "res", "inception"):
        out = {}
        model = build_model(arch)
        model.train()
        out.update(state_arrays(model, "init:"))
        g = torch.Generator().manual_seed(11)
        B = 8
        x = torch.rand(B, 3, 32, 32, generator=g)
        eps = torch.randn(B, 10, generator=g)
        probe_img = torch.randn(B, 3, 32, 32, generator=g)
        probe_mu = torch.randn(B, 10, generator=g)
        probe_lv = torch.randn(B, 10, generator=g)
        mu, logvar = model.encode(x)
        z = mu + eps * torch.exp(0.5 * logvar)
        rec = model.decode(z)
        scalar = (rec * probe_img).sum() + (mu * probe_mu).sum() + (logvar * probe_lv).sum()
        scalar.backward()
        out["x"], out["eps"] = npy(x), npy(eps)
        out["probe_img"], out["probe_mu"], out["probe_lv"] = npy(probe_img), npy(probe_mu), npy(probe_lv)
        out["mu"], out["logvar"], out["z"], out["rec"] = npy(mu), npy(logvar), npy(z), npy(rec)
        out["scalar"] = npy(scalar)
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["grad:" + k.replace(".", "/")] = npy(p.grad)
        out.update(state_arrays(model, "after_train_fwd:"))
        # eval-mode encode/decode on the updated running stats (evaluation/utils.py:52)
        model.eval()
        with torch.no_grad():
            mu_e, lv_e = model.encode(x)
            rec_e = model.decode(mu_e)
        out["eval_mu"], out["eval_logvar"], out["eval_rec"] = npy(mu_e), npy(lv_e), npy(rec_e)
        save(f"model_{arch}.npz", **out)


# ------------------------------------------------------------------------ solvers
class _DS:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


HP = dict(beta_kl=0.5, beta_rec=0.75, beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4)


class Recorder:
    """Records every RNG draw (torch.randn / torch.randn_like) and clip-norm result."""

    def __init__(self):
        self.draws, self.norms = [], []
        self._randn, self._randn_like = torch.randn, torch.randn_like
        self._clip = torch.nn.utils.clip_grad_norm_

    def __enter__(self):
        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.draws.append(t.clone())
            return t

        def randn_like(*a, **k):
            t = self._randn_like(*a, **k)
            self.draws.append(t.clone())
            return t

        def clip(*a, **k):
            n = self._clip(*a, **k)
            self.norms.append(n.clone())
            return n

        torch.randn, torch.randn_like = randn, randn_like
        torch.nn.utils.clip_grad_norm_ = clip
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like
        torch.nn.utils.clip_grad_norm_ = self._clip


def gen_steps(arch="conv", nsteps=2, B=8, N=1000, fname=None, loss_type="mse"):
    SingletonWriter().writer = None
    SingletonWriter().cur_iter = 0
    SingletonWriter().test_iter = N // B
    out = {}
    g = torch.Generator().manual_seed(21)
    xs = [torch.rand(B, 3, 32, 32, generator=g) for _ in range(nsteps)]
    out["hp"] = np.array([HP["beta_kl"], HP["beta_rec"], HP["beta_neg"], HP["gamma_r"], HP["clip"], HP["lr"], N], dtype=np.float64)
    for s, x in enumerate(xs):
        out[f"x{s}"] = npy(x)
    for name, cls in (("vae", VAESolver), ("tc", TCSovler), ("intro", IntroSolver), ("intro_tc", IntroTCSovler)):
        model = build_model(arch)
        model.train()
        if name == "vae":
            out.update(state_arrays(model, "init:"))
        opt_e = torch.optim.Adam(model.encoder.parameters(), lr=HP["lr"])
        opt_d = torch.optim.Adam(model.decoder.parameters(), lr=HP["lr"])
        kw = dict(dataset=_DS(N), model=model, batch_size=B, optimizer_e=opt_e, optimizer_d=opt_d,
                  recon_loss_type=loss_type, beta_kl=HP["beta_kl"], beta_rec=HP["beta_rec"],
                  device=torch.device("cpu"), use_amp=False, grad_scaler=None, writer=None,
                  test_iter=1000, clip=HP["clip"])
        if name.startswith("intro"):
            kw.update(beta_neg=HP["beta_neg"], gamma_r=HP["gamma_r"])
        solver = cls(**kw)
        # record the hook outputs in call order
        kl_calls, rec_calls = [], []
        kl_orig, rec_orig = solver.compute_kl_loss, solver.compute_rec_loss

        def kl_hook(*a, _f=kl_orig, **k):
            r = _f(*a, **k)
            kl_calls.append(r.detach().clone().reshape(-1))
            return r

        def rec_hook(*a, _f=rec_orig, **k):
            r = _f(*a, **k)
            rec_calls.append(r.detach().clone().reshape(-1))
            return r

        solver.compute_kl_loss, solver.compute_rec_loss = kl_hook, rec_hook
        torch.manual_seed(1234)
        for s, x in enumerate(xs):
            kl_calls.clear()
            rec_calls.clear()
            with Recorder() as rec:
                d = solver.train_step(x, s)
            p = f"{name}:s{s}:"
            out[p + "dict"] = np.array([d["loss_enc"], d["loss_dec"], d["loss_kl"], d["loss_rec"], d["L2"]], dtype=np.float64)
            for i, t in enumerate(rec.draws):
                out[p + f"draw{i}"] = npy(t)
            out[p + "norms"] = np.array([n.item() for n in rec.norms], dtype=np.float64)
            for i, t in enumerate(kl_calls):
                out[p + f"kl{i}"] = npy(t)
            for i, t in enumerate(rec_calls):
                out[p + f"rec{i}"] = npy(t)
            sd = model.state_dict()
            keys = list(sd.keys())
            out[p + "chk"] = np.array([[sd[k].double().sum().item(), sd[k].double().pow(2).sum().sqrt().item()] for k in keys], dtype=np.float64)
            if name == "vae" and s == 0:
                out["state_keys"] = np.array(keys)
        out.update(state_arrays(model, f"{name}:final:"))
    save(fname or f"steps_{arch}.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "models", "steps"]
    if "ops" in which:
        gen_ops()
    if "ops_c4" in which or not sys.argv[1:]:
        gen_ops_c4()
    if "models" in which:
        gen_models()
    if "steps" in which:
        gen_steps("conv")
        gen_steps("res", nsteps=1, fname="steps_res.npz")
        gen_steps("conv", nsteps=1, fname="steps_conv_bce.npz", loss_type="bce")
