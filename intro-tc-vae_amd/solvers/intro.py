"""IntroSolver: the Soft-Intro VAE step (/root/reference/solvers/intro.py:18-196) on HIP kernels.

Per call: 5 encoder and 8 decoder forward passes, two backward passes, two optimiser updates --
the schedule, the draw order of the six N(0,1) tensors and the frozen-half / stale-gradient
clip semantics follow the reference line by line (cited inline); the arithmetic is all in
libitcv_hip.so and the host reads the five returned scalars back in ONE transfer.
"""
from typing import Optional

import torch
from torch import Tensor

from hipvae import ddp
from hipvae.functional import ExpElboFn
from models import bn_groups
from ops import noise, reparameterize
from solvers.vae import VAESolver


class IntroSolver(VAESolver):
    def __init__(self, dataset, model, batch_size: int, optimizer_e, optimizer_d, recon_loss_type: str,
                 beta_kl: float, beta_rec: float, beta_neg: float, gamma_r: float, device: torch.device,
                 use_amp: bool, grad_scaler, writer=None, test_iter: int = 1000, clip: Optional[float] = None):
        super().__init__(dataset, model, batch_size, optimizer_e, optimizer_d, recon_loss_type, beta_kl, beta_rec,
                         device, use_amp, grad_scaler, writer, test_iter, clip)
        self.beta_neg = beta_neg
        self.gamma_r = gamma_r
        # Passes that the reference issues back to back on the same weights -- dec(noise) | dec(z), enc(rec) | enc(fake),
        # dec(z_rec) | dec(z_fake), in both phases -- run as ONE batched pass of 2B images with two BatchNorm groups
        # (models.bn_groups): 13 network passes become 7 launches-wise, every conv GEMM / weight gradient sees twice
        # the pixels, and the results are those of separate passes (per-pass batch statistics, same draw order, same
        # running-buffer order).  ``batch_passes = False`` issues the 13 passes one by one (the tests compare the two).
        self.batch_passes = True

    def _exp_elbo(self, rec_rows: Tensor, kl_rows: Tensor) -> Tensor:
        """intro.py:102-103  mean_j exp(-2 * scale * (rec_j + kl_j))."""
        while rec_rows.dim() > 1:
            rec_rows = rec_rows.sum(-1)
        if rec_rows.is_cuda and rec_rows.shape == kl_rows.shape and rec_rows.dim() == 1:
            return ExpElboFn.apply(rec_rows, kl_rows, -2 * self.scale)         # one launch each way
        return (-2 * self.scale * (rec_rows + kl_rows)).exp().mean()

    def _device_step(self, real: Tensor) -> Tensor:
        """Everything of intro.py:56-160 that runs on the device; returns the stats vector
        [loss_enc, loss_dec, loss_kl, loss_rec, expelbo_fake, lossD_fake_kl, norm_E, norm_D]."""
        if not self.batch_passes:
            return self._device_step_unbatched(real)
        model, scale = self.model, self.scale
        cat = torch.cat

        def two():
            return bn_groups(2)

        noise_batch = noise((real.size(0), model.zdim), self.device)               # intro.py:61
        # ================= update E (decoder frozen) ======================== intro.py:65-116
        self._set_trainable(encoder=True, decoder=False)
        real_mu, real_logvar = model.encode(real)
        z = reparameterize(real_mu, real_logvar)
        with two():                                    # fake = sample(noise) | rec = decoder(z)      intro.py:70,75
            fake, rec = model.decoder(cat([noise_batch, z])).chunk(2)
        loss_rec = self.compute_rec_loss(real, rec, reduction="mean")
        loss_e_real_kl = self.compute_kl_loss(z, real_mu, real_logvar, write=True)
        with two():                                    # model(rec.detach()) | model(fake.detach())   intro.py:81-82
            mu2, logvar2 = model.encode(cat([rec.detach(), fake.detach()]))
        (rec_mu, fake_mu), (rec_logvar, fake_logvar) = mu2.chunk(2), logvar2.chunk(2)
        z_rec = reparameterize(rec_mu, rec_logvar)
        z_fake = reparameterize(fake_mu, fake_logvar)
        with two():
            rec_rec, rec_fake = model.decoder(cat([z_rec, z_fake])).chunk(2)
        kl_rec = self.compute_kl_loss(z_rec, rec_mu, rec_logvar, reduce="none", beta=self.beta_neg)
        kl_fake = self.compute_kl_loss(z_fake, fake_mu, fake_logvar, reduce="none", beta=self.beta_neg)
        expelbo_rec = self._exp_elbo(self.compute_rec_loss(rec, rec_rec, reduction="none"), kl_rec)
        expelbo_fake = self._exp_elbo(self.compute_rec_loss(fake, rec_fake, reduction="none"), kl_fake)
        # scale * (loss_rec + loss_e_real_kl) + 0.25 * (expelbo_rec + expelbo_fake), intro.py:105-108
        loss_e = self._lincomb((scale, scale, 0.25, 0.25), loss_rec, loss_e_real_kl, expelbo_rec, expelbo_fake)
        finish_average = self._backward(loss_e, ("encoder",), defer_average=True)

        # ================= update D (encoder frozen) ======================== intro.py:118-160
        # The decoder-only pass that opens this phase depends neither on the encoder's gradients nor on its update: it
        # is issued while the encoder-gradient all-reduce is in flight (data-parallel runs), then the encoder update
        # completes.  Single-process: same kernels, same results, commuting order.
        self._set_trainable(encoder=False, decoder=True)
        with two():                                    # fake = sample(noise) | rec = decoder(z.detach())   intro.py:119-120
            fake, rec = model.decoder(cat([noise_batch, z.detach()])).chunk(2)
        finish_average()
        norm_e = self._clip()
        self._step("encoder")
        loss_rec = self.compute_rec_loss(real, rec, reduction="mean", write=True)
        with two():                                    # encode(rec) | encode(fake)                          intro.py:128-132
            mu2, logvar2 = model.encode(cat([rec, fake]))
        (rec_mu, fake_mu), (rec_logvar, fake_logvar) = mu2.chunk(2), logvar2.chunk(2)
        z_rec = reparameterize(rec_mu, rec_logvar)
        z_fake = reparameterize(fake_mu, fake_logvar)
        with two():                                    # decode(z_rec.detach()) | decode(z_fake.detach())   intro.py:133-134
            rec_rec, rec_fake = model.decoder(cat([z_rec.detach(), z_fake.detach()])).chunk(2)
        g = self.gamma_r * self.beta_rec
        loss_rec_rec = self.compute_rec_loss(rec.detach(), rec_rec, reduction="mean", beta=g)
        loss_fake_rec = self.compute_rec_loss(fake.detach(), rec_fake, reduction="mean", beta=g)
        loss_d_rec_kl = self.compute_kl_loss(z_rec, rec_mu, rec_logvar)
        loss_d_fake_kl = self.compute_kl_loss(z_fake, fake_mu, fake_logvar)
        # scale * (loss_rec + (kl_rec + kl_fake) * 0.5 + (rec_rec + fake_rec) * 0.5), intro.py:149-151
        loss_d = self._lincomb((scale, 0.5 * scale, 0.5 * scale, 0.5 * scale, 0.5 * scale), loss_rec, loss_d_rec_kl,
                               loss_d_fake_kl, loss_rec_rec, loss_fake_rec)
        self._backward(loss_d, ("decoder",))
        norm_d = self._clip()
        self._step("decoder")

        stats = torch.stack([loss_e.detach(), loss_d.detach(), loss_e_real_kl.detach(), loss_rec.detach(),
                             expelbo_fake.detach(), loss_d_fake_kl.detach()])
        ddp.mean_scalars_(stats)
        zero = stats.new_zeros(1)
        self._last_fake = fake.detach() if self.writer else None
        return torch.cat([stats, norm_e if norm_e is not None else zero, norm_d if norm_d is not None else zero])

    def _device_step_unbatched(self, real: Tensor) -> Tensor:
        """The step with its 13 network passes issued one by one, in the reference's statement order."""
        model, scale = self.model, self.scale
        noise_batch = noise((real.size(0), model.zdim), self.device)               # intro.py:61
        # ================= update E (decoder frozen) ======================== intro.py:65-116
        self._set_trainable(encoder=True, decoder=False)
        fake = model.sample(noise_batch)
        real_mu, real_logvar = model.encode(real)
        z = reparameterize(real_mu, real_logvar)
        rec = model.decoder(z)
        loss_rec = self.compute_rec_loss(real, rec, reduction="mean")
        loss_e_real_kl = self.compute_kl_loss(z, real_mu, real_logvar, write=True)
        rec_mu, rec_logvar, z_rec, rec_rec = model(rec.detach())
        fake_mu, fake_logvar, z_fake, rec_fake = model(fake.detach())
        kl_rec = self.compute_kl_loss(z_rec, rec_mu, rec_logvar, reduce="none", beta=self.beta_neg)
        kl_fake = self.compute_kl_loss(z_fake, fake_mu, fake_logvar, reduce="none", beta=self.beta_neg)
        expelbo_rec = self._exp_elbo(self.compute_rec_loss(rec, rec_rec, reduction="none"), kl_rec)
        expelbo_fake = self._exp_elbo(self.compute_rec_loss(fake, rec_fake, reduction="none"), kl_fake)
        # scale * (loss_rec + loss_e_real_kl) + 0.25 * (expelbo_rec + expelbo_fake), intro.py:105-108
        loss_e = self._lincomb((scale, scale, 0.25, 0.25), loss_rec, loss_e_real_kl, expelbo_rec, expelbo_fake)
        finish_average = self._backward(loss_e, ("encoder",), defer_average=True)

        # ================= update D (encoder frozen) ======================== intro.py:118-160
        # The two decoder-only forwards of this phase depend neither on the encoder's gradients nor on its update:
        # they are issued while the encoder-gradient all-reduce is in flight (data-parallel runs), then the encoder
        # update completes.  Single-process: same kernels, same results, commuting order.
        self._set_trainable(encoder=False, decoder=True)
        fake = model.sample(noise_batch)
        rec = model.decoder(z.detach())
        finish_average()
        norm_e = self._clip()
        self._step("encoder")
        loss_rec = self.compute_rec_loss(real, rec, reduction="mean", write=True)
        rec_mu, rec_logvar = model.encode(rec)
        z_rec = reparameterize(rec_mu, rec_logvar)
        fake_mu, fake_logvar = model.encode(fake)
        z_fake = reparameterize(fake_mu, fake_logvar)
        rec_rec = model.decode(z_rec.detach())
        rec_fake = model.decode(z_fake.detach())
        g = self.gamma_r * self.beta_rec
        loss_rec_rec = self.compute_rec_loss(rec.detach(), rec_rec, reduction="mean", beta=g)
        loss_fake_rec = self.compute_rec_loss(fake.detach(), rec_fake, reduction="mean", beta=g)
        loss_d_rec_kl = self.compute_kl_loss(z_rec, rec_mu, rec_logvar)
        loss_d_fake_kl = self.compute_kl_loss(z_fake, fake_mu, fake_logvar)
        # scale * (loss_rec + (kl_rec + kl_fake) * 0.5 + (rec_rec + fake_rec) * 0.5), intro.py:149-151
        loss_d = self._lincomb((scale, 0.5 * scale, 0.5 * scale, 0.5 * scale, 0.5 * scale), loss_rec, loss_d_rec_kl,
                               loss_d_fake_kl, loss_rec_rec, loss_fake_rec)
        self._backward(loss_d, ("decoder",))
        norm_d = self._clip()
        self._step("decoder")

        stats = torch.stack([loss_e.detach(), loss_d.detach(), loss_e_real_kl.detach(), loss_rec.detach(),
                             expelbo_fake.detach(), loss_d_fake_kl.detach()])
        ddp.mean_scalars_(stats)
        zero = stats.new_zeros(1)
        self._last_fake = fake.detach() if self.writer else None
        return torch.cat([stats, norm_e if norm_e is not None else zero, norm_d if norm_d is not None else zero])

    def train_step(self, batch: Tensor, cur_iter: int) -> dict:
        if batch.dim() == 3:
            batch = batch.unsqueeze(0)
        real = batch.to(self.device)
        # ================= one read-back, NaN check, logging ================ intro.py:162-196
        v_e, v_d, v_kl, v_rec, v_expf, v_dfkl, v_ne, v_nd = self._run(real).tolist()
        if v_e != v_e or v_d != v_d:
            raise RuntimeError
        if self.writer:
            self.write_scalars(cur_iter, losses=dict(r_loss=v_rec, kl_loss=v_kl, expelbo_f=v_expf),
                               diff_kl=v_dfkl - v_kl)
            if self.clip:
                self.writer.add_scalars("total_norm", {"E": v_ne, "D": v_nd}, global_step=cur_iter)
            self.writer.add_scalar("lossE", v_e, global_step=cur_iter)
            self.writer.add_scalar("lossD", v_d, global_step=cur_iter)
            self.write_gradient_norm(cur_iter)
            self.write_images(real, self._last_fake, cur_iter)
            self.write_disentanglemnt_scores(cur_iter)
            self.writer.flush()
        return {"loss_enc": v_e, "loss_dec": v_d, "loss_kl": v_kl, "loss_rec": v_rec,
                "L2": max(v_ne, v_nd) if self.clip else None}
