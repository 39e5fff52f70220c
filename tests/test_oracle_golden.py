"""Pins the CPU oracle (oracle/) to the golden vectors captured from the unmodified
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import latent_math as lm
from oracle.network import Net
from oracle.steps import Trainer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = dict(cdim=3, zdim=10, channels=(8, 16, 32), image_size=32)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def G():
    """ops.npz (tags a, b, c + reconstruction) and ops_c4.npz (tag d: the c4 size, B_glob=512, D=128, N=10000)."""
    both = {}
    for f in ("ops.npz", "ops_c4.npz"):
        z = np.load(os.path.join(GOLDEN, f))
        both.update({k: z[k] for k in z.files})
    return both


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_latent_forward(G, tag):
    B, D, N = (int(v) for v in G[f"{tag}_BDN"])
    z, mu, lv, eps = (T(G[f"{tag}_{k}"]) for k in ("z", "mu", "logvar", "eps"))
    close(lm.reparameterize(mu, lv, eps), G[f"{tag}_reparam"])
    close(lm.kl_rows(lv, mu), G[f"{tag}_kl_none"], rtol=1e-5)
    close(lm.kl(lv, mu, "sum"), G[f"{tag}_kl_sum"], rtol=1e-5)
    close(lm.kl(lv, mu, "mean"), G[f"{tag}_kl_mean"], rtol=1e-5)
    close(lm.log_importance_weights(B, N), G[f"{tag}_logiw"], rtol=1e-6)
    lp = lm.pairwise_live(z, mu, lv)
    pm, lq = lm.stratified(lp, N)
    close(pm, G[f"{tag}_mss_prodm"], rtol=2e-5)
    close(lq, G[f"{tag}_mss_logqz"], rtol=2e-5)
    pm, lq = lm.weighted(lp, N)
    close(pm, G[f"{tag}_mws_prodm"], rtol=2e-5)
    close(lq, G[f"{tag}_mws_logqz"], rtol=2e-5)
    close(lm.total_correlation(z, mu, lv, N, "none"), G[f"{tag}_tc_none"], rtol=1e-4, atol=1e-3)
    close(lm.total_correlation(z, mu, lv, N, "mean"), G[f"{tag}_tc_mean"], rtol=1e-4)
    mi, tc, dw = lm.decomposition(z, mu, lv, N)
    close(mi, G[f"{tag}_full_mi"], rtol=1e-4, atol=1e-3)
    close(tc, G[f"{tag}_full_tc"], rtol=1e-4, atol=1e-3)
    close(dw, G[f"{tag}_full_dwkl"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_latent_gradients(G, tag):
    B, D, N = (int(v) for v in G[f"{tag}_BDN"])
    for beta, bt in ((512.0, "512p0"), (0.5, "0p5")):
        z, mu, lv = (T(G[f"{tag}_{k}"]).clone().requires_grad_(True) for k in ("z", "mu", "logvar"))
        loss = lm.tc_kl(z, mu, lv, N, beta, "mean")
        loss.backward()
        close(loss, G[f"{tag}_tckl_b{bt}"], rtol=1e-4)
        scale = float(np.abs(G[f"{tag}_tckl_b{bt}_dlogvar"]).max())
        close(z.grad, G[f"{tag}_tckl_b{bt}_dz"], rtol=1e-4, atol=1e-5 * scale)
        close(mu.grad, G[f"{tag}_tckl_b{bt}_dmu"], rtol=1e-4, atol=1e-5 * scale)
        close(lv.grad, G[f"{tag}_tckl_b{bt}_dlogvar"], rtol=1e-4, atol=1e-5 * scale)
    z, mu, lv = (T(G[f"{tag}_{k}"]).clone().requires_grad_(True) for k in ("z", "mu", "logvar"))
    w = T(G[f"{tag}_tcw_w"])
    (w * lm.total_correlation(z, mu, lv, N, "none")).sum().backward()
    scale = float(np.abs(G[f"{tag}_tcw_dlogvar"]).max())
    close(z.grad, G[f"{tag}_tcw_dz"], rtol=1e-4, atol=1e-5 * scale)
    close(mu.grad, G[f"{tag}_tcw_dmu"], rtol=1e-4, atol=1e-5 * scale)
    close(lv.grad, G[f"{tag}_tcw_dlogvar"], rtol=1e-4, atol=1e-5 * scale)


def test_clamps_fire(G):
    """The fixtures exercise both clamps of ops.py:18,21."""
    for tag in "abcd":
        assert G[f"{tag}_frac_clamp50"][0] > 0.1
        assert G[f"{tag}_frac_varclamp"][0] > 0.0


def test_reconstruction(G):
    x, xr = T(G["rec_x"]), T(G["rec_xr"])
    for lt in ("mse", "l1", "bce"):
        for red in ("sum", "mean", "none"):
            close(lm.reconstruction_loss(x, xr, lt, red), G[f"rec_{lt}_{red}"], rtol=1e-5)
        xg = xr.clone().requires_grad_(True)
        (T(G[f"rec_{lt}_w"]) * lm.reconstruction_loss(x, xg, lt, "none")).sum().backward()
        close(xg.grad, G[f"rec_{lt}_dxr"], rtol=1e-5, atol=1e-6)
    # known-answer tests of the reference's tests/test_ops.py:10-43
    x0, x1 = torch.zeros(3), torch.tensor([1.0, 2.0, 4.0])
    assert lm.reconstruction_loss(x0, x1, "mse", "sum").item() == 21
    assert lm.reconstruction_loss(x0, x1, "mse", "mean").item() == 7
    assert list(lm.reconstruction_loss(x0, x1, "mse", "none").numpy()) == [1, 4, 16]
    assert lm.reconstruction_loss(x0, x1, "l1", "sum").item() == 7
    assert lm.reconstruction_loss(x0, x1, "l1", "mean").item() == pytest.approx(7 / 3)
    with pytest.raises(NotImplementedError):
        lm.reconstruction_loss(x0, x1, "huber", "sum")
    with pytest.raises(NotImplementedError):
        lm.reconstruction_loss(x0, x1, "mse", "avg")


def _state(npz, prefix):
    return {k[len(prefix):].replace("/", "."): T(npz[k]).clone() for k in npz.files if k.startswith(prefix)}


@pytest.mark.parametrize("arch", ["conv", "res", "inception"])
def test_network(arch):
    g = np.load(os.path.join(GOLDEN, f"model_{arch}.npz"))
    sd = _state(g, "init:")
    # fresh-encoder BatchNorm state after the reference's dummy forward (models.py:229-238)
    assert float(sd["encoder.main.1.running_var"][0]) == pytest.approx(0.9)
    assert int(sd["encoder.main.1.num_batches_tracked"]) == 1
    assert int(sd["decoder.main.res_in_4.bn1.num_batches_tracked"] if arch != "inception"
               else sd["decoder.main.res_in_4.branch_0.batch_norm.num_batches_tracked"]) == 0
    net = Net(arch, state=sd, **TINY)
    for k in net.param_keys("encoder") + net.param_keys("decoder"):
        sd[k].requires_grad_(True)
    x, eps = T(g["x"]), T(g["eps"])
    mu, lv = net.encode(x)
    z = lm.reparameterize(mu, lv, eps)
    rec = net.decode(z)
    close(mu, g["mu"], rtol=1e-4, atol=1e-5)
    close(lv, g["logvar"], rtol=1e-4, atol=1e-5)
    close(rec, g["rec"], rtol=1e-4, atol=1e-5)
    s = (rec * T(g["probe_img"])).sum() + (mu * T(g["probe_mu"])).sum() + (lv * T(g["probe_lv"])).sum()
    close(s, g["scalar"], rtol=1e-4)
    s.backward()
    n = 0
    for k in g.files:
        if k.startswith("grad:"):
            key = k[5:].replace("/", ".")
            ref = g[k]
            close(sd[key].grad, ref, rtol=1e-3, atol=1e-4 * max(1e-6, float(np.abs(ref).max())))
            n += 1
    assert n > 10
    unused = [k for k in net.param_keys("encoder") + net.param_keys("decoder") if sd[k].grad is None]
    if arch == "conv":
        assert unused and all("conv_expand" in k for k in unused)
    after = _state(g, "after_train_fwd:")
    for k, v in after.items():
        if not Net.is_param(k):
            close(sd[k].detach(), v.numpy(), rtol=1e-4, atol=1e-6)
    net.train = False
    with torch.no_grad():
        mu_e, lv_e = net.encode(x)
        rec_e = net.decode(mu_e)
    close(mu_e, g["eval_mu"], rtol=1e-4, atol=1e-5)
    close(rec_e, g["eval_rec"], rtol=1e-4, atol=1e-5)


def _run_steps(fname, arch, loss_type, names, nsteps):
    g = np.load(os.path.join(GOLDEN, fname))
    hp = g["hp"]
    keys = [str(k) for k in g["state_keys"]]
    for name in names:
        sd = _state(g, "init:")
        net = Net(arch, state=sd, **TINY)
        tr = Trainer(name, net, dataset_size=int(hp[6]), recon_loss_type=loss_type, beta_kl=hp[0],
                     beta_rec=hp[1], beta_neg=hp[2], gamma_r=hp[3], clip=hp[4], lr=hp[5])
        for s in range(nsteps):
            p = f"{name}:s{s}:"
            ndraw = len([k for k in g.files if k.startswith(p + "draw")])
            draws = [T(g[p + f"draw{i}"]) for i in range(ndraw)]
            d = tr.step(T(g[f"x{s}"]), draws)
            got = np.array([d["loss_enc"], d["loss_dec"], d["loss_kl"], d["loss_rec"], d["L2"]])
            np.testing.assert_allclose(got, g[p + "dict"], rtol=2e-4, err_msg=p)
            np.testing.assert_allclose(np.array(tr.trace["norms"], dtype=np.float64), g[p + "norms"], rtol=2e-4)
            for i, t in enumerate(tr.trace["kl"]):
                ref = g[p + f"kl{i}"]
                close(t, ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))
            for i, t in enumerate(tr.trace["rec"]):
                close(t, g[p + f"rec{i}"], rtol=2e-4)
            chk = np.array([[sd[k].double().sum().item(), sd[k].double().pow(2).sum().sqrt().item()] for k in keys])
            np.testing.assert_allclose(chk[:, 1], g[p + "chk"][:, 1], rtol=1e-4, atol=1e-6, err_msg=p)
        fin = _state(g, f"{name}:final:")
        worst = 0.0
        for k, v in fin.items():
            if v.dtype.is_floating_point:
                worst = max(worst, float((sd[k].detach() - v).abs().max()))
        # Adam moves every weight by ~lr per step; agreement well below that
        assert worst < 0.25 * hp[5] * nsteps + 1e-6, (name, worst)


def test_steps_conv():
    _run_steps("steps_conv.npz", "conv", "mse", ("vae", "tc", "intro", "intro_tc"), 2)


def test_steps_res():
    _run_steps("steps_res.npz", "res", "mse", ("vae", "tc", "intro", "intro_tc"), 1)


def test_steps_bce():
    _run_steps("steps_conv_bce.npz", "conv", "bce", ("vae", "intro_tc"), 1)
