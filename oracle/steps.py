"""Oracle restatement of the reference's solvers (test infrastructure, see oracle/__init__).

``Trainer`` restates VAESolver / TCSovler / IntroSolver / IntroTCSovler ``train_step``
(/root/reference/solvers/vae.py:89-136, solvers/intro.py:56-196, solvers/tc.py:58-89,
solvers/intro_tc.py:7-17) over the functional network of oracle.network, together with
torch.optim.Adam's default update and torch.nn.utils.clip_grad_norm_ as called by the
reference (including the stale-gradient behaviour of solvers/intro.py:113-115,157-159:
the norm runs over every parameter whose ``.grad`` is not None, frozen half included).

Random draws are explicit inputs, in the reference's draw order:
  vae / tc  : [eps]                                   (models.py:337 -> ops.py:184)
  intro(-tc): [noise, eps_real, eps_rec, eps_fake,    (intro.py:61,74,81,82)
               eps_rec_D, eps_fake_D]                 (intro.py:129,132)
"""
import torch

from . import latent_math as lm
from .network import Net


class Trainer:
    def __init__(self, solver, net: Net, dataset_size, recon_loss_type="mse", beta_kl=1.0,
                 beta_rec=1.0, beta_neg=1.0, gamma_r=1e-8, clip=None, lr=2e-4,
                 adam_betas=(0.9, 0.999), adam_eps=1e-8):
        assert solver in ("vae", "tc", "intro", "intro_tc")
        self.solver, self.net, self.n = solver, net, dataset_size
        self.loss_type = recon_loss_type
        self.beta_kl, self.beta_rec, self.beta_neg, self.gamma_r = beta_kl, beta_rec, beta_neg, gamma_r
        self.clip, self.lr, self.betas, self.adam_eps = clip, lr, adam_betas, adam_eps
        self.scale = 1.0 / (net.cdim * net.image_size ** 2)          # solvers/vae.py:61
        self.keys = {p: net.param_keys(p) for p in ("encoder", "decoder")}
        for ks in self.keys.values():
            for k in ks:
                net.sd[k].requires_grad_(True)
        self.grads = {}                       # key -> Tensor | absent (== .grad is None)
        self.adam = {}                        # key -> [step, exp_avg, exp_avg_sq]
        self.trace = {}                       # last step's intermediates (for tests)

    # ---- overridable loss hooks (solvers/vae.py:63-87, solvers/tc.py:58-89) -------------
    def kl_loss(self, z, mu, logvar, reduce="mean", beta=None):
        beta = self.beta_kl if beta is None else beta
        if self.solver in ("tc", "intro_tc"):
            out = lm.tc_kl(z, mu, logvar, self.n, beta, reduce)
        else:
            out = beta * lm.kl(logvar, mu, reduce)
        self.trace.setdefault("kl", []).append(out.detach().reshape(-1).clone())
        return out

    def rec_loss(self, x, recon, reduction="sum", beta=None):
        beta = self.beta_rec if beta is None else beta
        out = beta * lm.reconstruction_loss(x, recon, self.loss_type, reduction)
        self.trace.setdefault("rec", []).append(out.detach().reshape(-1).clone())
        return out

    # ---- optimiser pieces ------------------------------------------------------------
    def _backward(self, loss, parts):
        """optimizer_<part>.zero_grad() for each part, then loss.backward() restricted to
        those halves (the other half is frozen via requires_grad=False in the reference)."""
        ks = [k for part in parts for k in self.keys[part]]
        gs = torch.autograd.grad(loss, [self.net.sd[k] for k in ks], allow_unused=True)
        for k, g in zip(ks, gs):
            if g is None:
                self.grads.pop(k, None)
            else:
                self.grads[k] = g.detach().clone()

    def _clip(self):
        """torch.nn.utils.clip_grad_norm_(model.parameters(), clip) over non-None grads."""
        order = [k for k in self.keys["encoder"] + self.keys["decoder"] if k in self.grads]
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(self.grads[k]) for k in order]))
        coef = torch.clamp(self.clip / (total + 1e-6), max=1.0)
        for k in order:
            self.grads[k].mul_(coef)
        return float(total)

    def _adam(self, part):
        """torch.optim.Adam defaults (amsgrad False, weight_decay 0), per-parameter state."""
        b1, b2 = self.betas
        with torch.no_grad():
            for k in self.keys[part]:
                g = self.grads.get(k)
                if g is None:
                    continue
                p = self.net.sd[k]
                st = self.adam.setdefault(k, [0, torch.zeros_like(p), torch.zeros_like(p)])
                st[0] += 1
                st[1].lerp_(g, 1.0 - b1)
                st[2].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                bc1 = 1.0 - b1 ** st[0]
                bc2 = 1.0 - b2 ** st[0]
                denom = (st[2].sqrt() / (bc2 ** 0.5)).add_(self.adam_eps)
                p.addcdiv_(st[1], denom, value=-self.lr / bc1)

    # ---- steps -----------------------------------------------------------------------
    def step(self, batch, draws):
        self.trace = {}
        if batch.dim() == 3:
            batch = batch.unsqueeze(0)
        if self.solver in ("vae", "tc"):
            return self._step_vae(batch, draws)
        return self._step_intro(batch, draws)

    def _step_vae(self, real, draws):
        """solvers/vae.py:89-136."""
        net = self.net
        mu, logvar = net.encode(real)
        z = lm.reparameterize(mu, logvar, draws[0])
        rec = net.decode(z)
        loss_rec = self.rec_loss(real, rec, "mean")
        loss_kl = self.kl_loss(z, mu, logvar)
        loss = self.scale * (loss_rec + loss_kl)
        self._backward(loss, ("decoder", "encoder"))
        norm = self._clip() if self.clip else None
        self._adam("encoder")
        self._adam("decoder")
        if torch.isnan(loss):
            raise RuntimeError
        self.trace.update(norms=[norm], last_rec=rec.detach())
        # the reference returns an unbound total_norm when clip is falsy (vae.py:135)
        return {"loss_enc": float(loss.detach()), "loss_dec": float(loss.detach()), "loss_kl": float(loss_kl.detach()),
                "loss_rec": float(loss_rec.detach()), "L2": norm}

    def _step_intro(self, real, draws):
        """solvers/intro.py:56-196 (IntroSolver) / solvers/intro_tc.py (TC KL hook)."""
        net, scale = self.net, self.scale
        noise, e_real, e_rec, e_fake, e_rec_d, e_fake_d = draws
        # ---- update E (decoder frozen) -------------------------------------------------
        fake = net.decode(noise)
        real_mu, real_logvar = net.encode(real)
        z = lm.reparameterize(real_mu, real_logvar, e_real)
        rec = net.decode(z)
        loss_rec = self.rec_loss(real, rec, "mean")
        loss_e_real_kl = self.kl_loss(z, real_mu, real_logvar)
        rec_mu, rec_logvar = net.encode(rec.detach())
        z_rec = lm.reparameterize(rec_mu, rec_logvar, e_rec)
        rec_rec = net.decode(z_rec)
        fake_mu, fake_logvar = net.encode(fake.detach())
        z_fake = lm.reparameterize(fake_mu, fake_logvar, e_fake)
        rec_fake = net.decode(z_fake)
        kl_rec = self.kl_loss(z_rec, rec_mu, rec_logvar, "none", self.beta_neg)
        kl_fake = self.kl_loss(z_fake, fake_mu, fake_logvar, "none", self.beta_neg)
        rr_e = self.rec_loss(rec, rec_rec, "none")
        rf_e = self.rec_loss(fake, rec_fake, "none")
        expelbo_rec = (-2.0 * scale * (rr_e + kl_rec)).exp().mean()
        expelbo_fake = (-2.0 * scale * (rf_e + kl_fake)).exp().mean()
        loss_e = scale * (loss_rec + loss_e_real_kl) + 0.25 * (expelbo_rec + expelbo_fake)
        self._backward(loss_e, ("encoder",))
        norm_e = self._clip() if self.clip else None
        self._adam("encoder")
        # ---- update D (encoder frozen) -------------------------------------------------
        fake = net.decode(noise)
        rec = net.decode(z.detach())
        loss_rec = self.rec_loss(real, rec, "mean")
        rec_mu, rec_logvar = net.encode(rec)
        z_rec = lm.reparameterize(rec_mu, rec_logvar, e_rec_d)
        fake_mu, fake_logvar = net.encode(fake)
        z_fake = lm.reparameterize(fake_mu, fake_logvar, e_fake_d)
        rec_rec = net.decode(z_rec.detach())
        rec_fake = net.decode(z_fake.detach())
        g = self.gamma_r * self.beta_rec
        loss_rec_rec = self.rec_loss(rec.detach(), rec_rec, "mean", g)
        loss_fake_rec = self.rec_loss(fake.detach(), rec_fake, "mean", g)
        kl_rec_d = self.kl_loss(z_rec, rec_mu, rec_logvar)
        kl_fake_d = self.kl_loss(z_fake, fake_mu, fake_logvar)
        loss_d = scale * (loss_rec + 0.5 * (kl_rec_d + kl_fake_d) + 0.5 * (loss_rec_rec + loss_fake_rec))
        self._backward(loss_d, ("decoder",))
        norm_d = self._clip() if self.clip else None
        self._adam("decoder")
        if torch.isnan(loss_d) or torch.isnan(loss_e):
            raise RuntimeError
        self.trace.update(norms=[norm_e, norm_d], expelbo=[float(expelbo_rec.detach()), float(expelbo_fake.detach())],
                          rec=self.trace.get("rec"), last_rec=rec.detach(), last_fake=fake.detach())
        return {"loss_enc": float(loss_e.detach()), "loss_dec": float(loss_d.detach()), "loss_kl": float(loss_e_real_kl.detach()),
                "loss_rec": float(loss_rec.detach()), "L2": max(norm_e, norm_d)}
