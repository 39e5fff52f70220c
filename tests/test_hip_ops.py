"""GPU parity tests of every C-ABI kernel family (through hipvae.functional -> ctypes ->
libitcv_hip.so) against plain PyTorch fp64/fp32 CPU references and the pinned oracle.

Tolerances (fp32 path, stated per check): convolutions / linear 2e-5 of the output scale
(fp32 fma chains vs an fp64 reference), statistics 1e-5, latent TC terms 1e-4 relative
(the bar north_star sets for ELBO / KL terms)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def HF():
    from hipvae import functional
    return functional


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CONV_CASES = [
    # B, Ci, H, W, Co, KS, up2
    (2, 3, 16, 16, 8, 5, False),      # stem-like, K = 75 (tail in K)
    (3, 8, 8, 8, 16, 3, False),
    (2, 16, 12, 20, 40, 3, False),    # non power-of-two spatial, M tail
    (4, 32, 4, 4, 32, 3, False),      # tiny spatial: N = 64 < tile
    (2, 64, 16, 16, 3, 5, False),     # predict-like: Co = 3
    (2, 24, 8, 8, 130, 3, False),     # two M tiles with tail
    (5, 96, 1, 1, 70, 1, False),      # linear as 1x1
    (2, 8, 16, 16, 8, 3, True),       # fused nearest upsample
    (1, 12, 8, 8, 20, 1, True),
    (2, 512, 4, 4, 64, 3, False),     # deep K = 4608 -> split-K path
]


@pytest.fixture(params=["fp32", "bf16x6", "bf16x3"])
def math_mode(request, HF):
    HF.set_conv_math(request.param)
    yield request.param
    HF.set_conv_math("fp32")


SPLIT_CASES = [  # shapes the split-bf16 kernels cover (Ci % 32 == 0, Co > 32, KS in {1,3}; wgrad also W % 8 == 0)
    (2, 32, 8, 8, 40, 3, False), (3, 64, 16, 16, 64, 3, False), (2, 96, 12, 20, 130, 3, False),
    (3, 128, 8, 16, 96, 3, False), (2, 64, 8, 8, 64, 1, False), (2, 256, 8, 8, 130, 3, False), (3, 32, 24, 8, 64, 3, True),
    (4, 128, 4, 4, 256, 3, False), (5, 64, 1, 1, 70, 1, False), (2, 64, 16, 16, 48, 3, True), (2, 512, 4, 4, 64, 3, False),
    # 128- and 256-wide images (BASELINE configs[2] / [4]): the weight gradient walks 64-column segments of a row
    (2, 32, 8, 128, 40, 3, False), (1, 64, 4, 256, 64, 3, False), (1, 32, 16, 256, 72, 3, True), (3, 64, 2, 128, 128, 3, True),
    # weight-heavy 4x4 layers at the benchmark's pixel counts: split-K over the 9 * Ci / 32 K-tiles
    (32, 512, 4, 4, 256, 3, False), (64, 256, 4, 4, 520, 3, False), (32, 256, 4, 4, 256, 3, True),
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_split_bf16_modes(HF, math_mode, case):
    """Forward and data-gradient in every conv arithmetic against fp64: exact fp32 MFMA and the
    fp32-class bf16x6 split at 2e-5 of the output scale, bf16x3 at 5e-5 (2^-16 per product)."""
    B, Ci, H, W, Co, KS, up2 = case
    assert HF.lib.itcv_conv2d_bf16s_supported(Ci, Co, KS)
    g = torch.Generator().manual_seed(sum(case[:6]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g)
    w = torch.randn(Co, Ci, KS, KS, generator=g) / (Ci * KS * KS) ** 0.5
    dy = torch.randn(B, Co, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up2 else xr
    yr = F.conv2d(xin, wr, padding=KS // 2)
    yr.backward(dy.double())
    xd, wd = x.to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True)
    y = HF.Conv2dFn.apply(xd, wd, None, up2)
    y.backward(dy.to(dev()))
    tol = 5e-5 if math_mode == "bf16x3" else 2e-5
    assert rel_err(y, yr) < tol and rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_backward(HF, case):
    B, Ci, H, W, Co, KS, up2 = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g)
    w = torch.randn(Co, Ci, KS, KS, generator=g) / (Ci * KS * KS) ** 0.5
    b = torch.randn(Co, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up2 else xr
    yr = F.conv2d(xin, wr, br, padding=KS // 2)
    yr.backward(dy.double())
    xd, wd, bd = (t.to(dev()).requires_grad_(True) for t in (x, w, b))
    y = HF.Conv2dFn.apply(xd, wd, bd, up2)
    y.backward(dy.to(dev()))
    torch.cuda.synchronize()
    assert rel_err(y, yr) < 2e-5
    assert rel_err(xd.grad, xr.grad) < 2e-5
    assert rel_err(wd.grad, wr.grad) < 2e-5
    assert rel_err(bd.grad, br.grad) < 2e-5


def test_conv_full_size_layer(HF):
    """One full-resolution layer of the 64x64 configuration (64->64 @64x64, 3x3; B=16 keeps the
    fp64 CPU reference to a few seconds) -- many N tiles, K = 65536 in the weight gradient."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 64, 64, 64, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    dy = torch.randn(16, 64, 64, 64, generator=g)
    xd, wd = x.to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True)
    y = HF.Conv2dFn.apply(xd, wd, None, False)
    y.backward(dy.to(dev()))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    yr.backward(dy.double())
    assert rel_err(y, yr) < 2e-5
    assert rel_err(xd.grad, xr.grad) < 2e-5
    assert rel_err(wd.grad, wr.grad) < 5e-5     # K = 262144-term sums in fp32 slabs


@pytest.mark.parametrize("shape", [(64, 512, 40), (5, 96, 70), (64, 8192, 256), (64, 128, 8192), (3, 33, 130)])
def test_linear(HF, shape):
    """nn.Linear on the skinny fp32 GEMMs: ragged tiles, both fc shapes of the c2 model (split-K and not)."""
    B, K, N = shape
    g = torch.Generator().manual_seed(3 + B + N)
    x = torch.randn(B, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(B, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    F.linear(xr, wr, br).backward(dy.double())
    xd, wd, bd = (t.to(dev()).requires_grad_(True) for t in (x, w, b))
    y = HF.LinearFn.apply(xd, wd, bd)
    y.backward(dy.to(dev()))
    assert rel_err(y, F.linear(xr, wr, br)) < 2e-5
    assert rel_err(xd.grad, xr.grad) < 2e-5
    assert rel_err(wd.grad, wr.grad) < 2e-5
    assert rel_err(bd.grad, br.grad) < 2e-5


@pytest.mark.parametrize("pool,skip,slope", [(False, False, 0.2), (True, False, 0.2), (False, True, 0.2),
                                              (True, True, 0.2), (False, False, 1.0)])
@pytest.mark.parametrize("shape", [(4, 6, 8, 8), (3, 5, 4, 12), (16, 32, 16, 16)])
def test_batchnorm_act(HF, shape, pool, skip, slope):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    sk = torch.randn(B, C, H, W, generator=g) if skip else None
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    oshape = (B, C, H // 2, W // 2) if pool else shape
    dy = torch.randn(*oshape, generator=g)
    eps = 1e-4
    # fp64 reference
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    skr = sk.double().requires_grad_(True) if skip else None
    rmr, rvr = rm.double().clone(), rv.double().clone()
    u = F.batch_norm(xr, rmr, rvr, gr, br, training=True, momentum=0.1, eps=eps)
    if skip:
        u = u + skr
    yr = F.leaky_relu(u, slope)
    if pool:
        yr = F.avg_pool2d(yr, 2)
    yr.backward(dy.double())
    d = dev()
    xd, gd, bd = (t.to(d).requires_grad_(True) for t in (x, gamma, beta))
    skd = sk.to(d).requires_grad_(True) if skip else None
    rmd, rvd, nbt = rm.to(d), rv.to(d), torch.zeros((), dtype=torch.long, device=d)
    y = HF.BnActFn.apply(xd, gd, bd, skd, rmd, rvd, nbt, eps, 0.1, slope, pool, True, None)
    y.backward(dy.to(d))
    assert rel_err(y, yr) < 1e-5
    assert rel_err(rmd, rmr) < 1e-5 and rel_err(rvd, rvr) < 1e-5 and int(nbt) == 1
    assert rel_err(xd.grad, xr.grad) < 2e-5
    assert rel_err(gd.grad, gr.grad) < 2e-5 and rel_err(bd.grad, br.grad) < 2e-5
    if skip:
        assert rel_err(skd.grad, skr.grad) < 1e-6
    # eval mode uses the running statistics
    ye = HF.BnActFn.apply(x.to(d), gamma.to(d), beta.to(d), None, rmd, rvd, nbt, eps, 0.1, slope, pool, False, None)
    ue = F.leaky_relu(F.batch_norm(x.double(), rmd.double().cpu(), rvd.double().cpu(), gamma.double(), beta.double(),
                                   training=False, eps=eps), slope)
    assert rel_err(ye, F.avg_pool2d(ue, 2) if pool else ue) < 1e-5


def test_bn_upsampled_gradient_mode(HF):
    """itcv_bn_act_bwd_* with up2=1 (gradient arriving at x2 resolution) == upsample adjoint + plain."""
    from hipvae.abi import call, lib, ptr, stream
    B, C, H, W = 3, 4, 6, 8
    g = torch.Generator().manual_seed(9)
    d = dev()
    x = torch.randn(B, C, H, W, generator=g).to(d)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(d), torch.randn(C, generator=g).to(d)
    mean, rstd = x.mean((0, 2, 3)), 1.0 / (x.var((0, 2, 3), unbiased=False) + 1e-4).sqrt()
    dy_hi = torch.randn(B, C, 2 * H, 2 * W, generator=g).to(d)
    dy_lo = dy_hi.view(B, C, H, 2, W, 2).sum((3, 5)).contiguous()
    outs = []
    for dy, up2 in ((dy_hi, 1), (dy_lo, 0)):
        nws = lib.itcv_bn_workspace(B, C, H * W)
        ws = torch.empty(nws, dtype=torch.uint8, device=d)
        sums = torch.empty(2 * C, dtype=torch.float64, device=d)
        call("itcv_bn_act_bwd_reduce", ptr(x), ptr(dy), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), None, ptr(sums),
             None, None, 0, B, C, H, W, 0.2, 0, up2, ptr(ws), nws, stream())
        dx = torch.empty_like(x)
        call("itcv_bn_act_bwd_apply", ptr(x), ptr(dy), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), None, ptr(sums),
             None, float(B * H * W), ptr(dx), None, None, None, 0, B, C, H, W, 0.2, 0, up2, None, 0, 0, stream())
        outs.append((sums.clone(), dx))
    assert rel_err(outs[0][0], outs[1][0]) < 1e-6
    assert rel_err(outs[0][1], outs[1][1]) < 1e-5


def test_pointwise_and_resampling(HF):
    g = torch.Generator().manual_seed(2)
    for shape in ((3, 5, 8, 12), (2, 3, 4, 6), (40, 64, 16, 16)):   # W % 4 == 0 takes the 4-wide upsample adjoint
        x = torch.randn(*shape, generator=g)
        for fn, ref in ((lambda t: HF.LeakyReluFn.apply(t, 0.2), lambda t: F.leaky_relu(t, 0.2)),
                        (HF.SigmoidFn.apply, torch.sigmoid),
                        (HF.AvgPool2Fn.apply, lambda t: F.avg_pool2d(t, 2)),
                        (HF.Upsample2Fn.apply, lambda t: F.interpolate(t, scale_factor=2, mode="nearest"))):
            xr = x.double().requires_grad_(True)
            yr = ref(xr)
            dy = torch.randn(*yr.shape, generator=g)
            yr.backward(dy.double())
            xd = x.to(dev()).requires_grad_(True)
            y = fn(xd)
            y.backward(dy.to(dev()))
            assert rel_err(y, yr) < 1e-6 and rel_err(xd.grad, xr.grad) < 1e-6
    a, b = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert rel_err(HF.AddFn.apply(a.to(dev()), b.to(dev())), a + b) == 0


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_latent_against_golden(HF, tag):
    """Reference-generated vectors (tests/golden/ops.npz; tag d = ops_c4.npz, the c4 size B=512, D=128, N=10000):
    KL, reparameterise, MSS/MWS, TC, the full decomposition and the gradients of (beta-1)TC + KL, with both clamps
    of ops.py:18,21 firing."""
    import ops
    G = np.load(os.path.join(GOLDEN, "ops_c4.npz" if tag == "d" else "ops.npz"))
    B, D, N = (int(v) for v in G[f"{tag}_BDN"])
    d = dev()
    z, mu, lv, eps = (torch.from_numpy(G[f"{tag}_{k}"]).to(d) for k in ("z", "mu", "logvar", "eps"))
    T = lambda k: torch.from_numpy(G[f"{tag}_{k}"])  # noqa: E731
    assert rel_err(HF.ReparamFn.apply(mu, lv, eps), T("reparam")) < 1e-6
    assert rel_err(ops.kl_no_reduce(lv, mu), T("kl_none")) < 1e-5
    assert rel_err(ops.kl_divergence(lv, mu, "mean"), T("kl_mean")) < 1e-5
    pm, lq = ops.tc_components(z, mu, lv, N)
    assert rel_err(pm, T("mss_prodm")) < 1e-4 and rel_err(lq, T("mss_logqz")) < 1e-4
    pm, lq = ops.tc_components(z, mu, lv, N, weighted=True)
    assert rel_err(pm, T("mws_prodm")) < 1e-4 and rel_err(lq, T("mws_logqz")) < 1e-4
    assert rel_err(ops.total_correlation(z, mu, lv, N, "none"), T("tc_none")) < 1e-4
    assert rel_err(ops.total_correlation(z, mu, lv, N, "mean"), T("tc_mean")) < 1e-4
    mi, tc, dw = ops.tc_decomposition(z, mu, lv, N)
    assert rel_err(mi, T("full_mi")) < 1e-4 and rel_err(tc, T("full_tc")) < 1e-4 and rel_err(dw, T("full_dwkl")) < 1e-4
    for beta, bt in ((512.0, "512p0"), (0.5, "0p5")):
        zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, lv))
        loss = (beta - 1.0) * ops.total_correlation(zz, mm, ll, N, "mean") + ops.kl_divergence(ll, mm, "mean")
        loss.backward()
        assert rel_err(loss, T(f"tckl_b{bt}")) < 1e-4
        assert rel_err(zz.grad, T(f"tckl_b{bt}_dz")) < 1e-4
        assert rel_err(mm.grad, T(f"tckl_b{bt}_dmu")) < 1e-4
        assert rel_err(ll.grad, T(f"tckl_b{bt}_dlogvar")) < 1e-4
    zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, lv))
    w = T("tcw_w").to(d)
    (w * ops.total_correlation(zz, mm, ll, N, "none")).sum().backward()
    assert rel_err(zz.grad, T("tcw_dz")) < 1e-4
    assert rel_err(mm.grad, T("tcw_dmu")) < 1e-4
    assert rel_err(ll.grad, T("tcw_dlogvar")) < 1e-4


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_named_density_and_sampling_functions_against_golden(HF, tag):
    """The reference's building blocks under their own names (ops.py:15-29, 92-123), composed the way the reference
    composes them (ops.py:80-89, solvers/tc.py:104-121): the materialised [B,B,D] log density against the oracle's,
    the two samplers / the total correlation / the full decomposition against the reference-generated vectors, and the
    gradients of (beta-1) TC + KL built from the named pieces against the reference's autograd.  1e-4 relative."""
    import ops
    from oracle import latent_math as OM
    G = np.load(os.path.join(GOLDEN, "ops.npz"))
    B, D, N = (int(v) for v in G[f"{tag}_BDN"])
    d = dev()
    zc, muc, lvc = (torch.from_numpy(G[f"{tag}_{k}"]) for k in ("z", "mu", "logvar"))
    z, mu, lv = zc.to(d), muc.to(d), lvc.to(d)
    T = lambda k: torch.from_numpy(G[f"{tag}_{k}"])  # noqa: E731
    lp = ops.gaussian_log_density_torch(z.unsqueeze(1), mu.unsqueeze(0), lv.unsqueeze(1))
    assert lp.shape == (B, B, D)
    ref_lp = OM.log_density_clamped_var(zc.unsqueeze(1), muc.unsqueeze(0), lvc.unsqueeze(1))
    assert float((lp.cpu() - ref_lp).abs().max()) < 1e-4 * 50          # values span [-50, ~5]
    assert abs(float((lp <= -50).float().mean()) - float(T("frac_clamp50")[0])) < 1e-3
    pm, lq = ops.minibatch_stratified_sampling(lp, B, N)
    assert rel_err(pm, T("mss_prodm")) < 1e-4 and rel_err(lq, T("mss_logqz")) < 1e-4
    pm, lq = ops.minibatch_weighted_sampling(lp, B, N)
    assert rel_err(pm, T("mws_prodm")) < 1e-4 and rel_err(lq, T("mws_logqz")) < 1e-4
    # solvers/tc.py:104-121: plain density, variance of component i
    logq_cx = ops.gaussian_log_density(z, mu, lv).sum(1)
    logpz = ops.gaussian_log_density(z, torch.zeros_like(z), torch.zeros_like(z)).sum(1)
    lp2 = ops.gaussian_log_density(z.unsqueeze(1), mu.unsqueeze(0), lv.unsqueeze(0))
    pm2, lq2 = ops.minibatch_stratified_sampling(lp2, B, N)
    assert rel_err(logq_cx, T("full_logq_cx")) < 1e-4 and rel_err(logpz, T("full_logpz")) < 1e-4
    assert rel_err(pm2, T("full_prodm")) < 1e-4 and rel_err(lq2, T("full_logqz")) < 1e-4
    assert rel_err(logq_cx - lq2, T("full_mi")) < 1e-4 and rel_err(pm2 - logpz, T("full_dwkl")) < 1e-4
    # gradients through the named pieces == the reference's autograd through its own
    for beta, bt in ((512.0, "512p0"), (0.5, "0p5")):
        zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, lv))
        lpg = ops.gaussian_log_density_torch(zz.unsqueeze(1), mm.unsqueeze(0), ll.unsqueeze(1))
        a, b = ops.minibatch_stratified_sampling(lpg, B, N)
        loss = (beta - 1.0) * (b - a).mean() + ops.kl_divergence(ll, mm, "mean")
        loss.backward()
        assert rel_err(loss, T(f"tckl_b{bt}")) < 1e-4
        assert rel_err(zz.grad, T(f"tckl_b{bt}_dz")) < 1e-4
        assert rel_err(mm.grad, T(f"tckl_b{bt}_dmu")) < 1e-4
        assert rel_err(ll.grad, T(f"tckl_b{bt}_dlogvar")) < 1e-4
    # the plain density's gradient and the weighted sampler's, against torch autograd on the oracle's restatement
    zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, lv))
    a, b = ops.minibatch_weighted_sampling(ops.gaussian_log_density(zz.unsqueeze(1), mm.unsqueeze(0), ll.unsqueeze(0)), B, N)
    w = torch.linspace(-1.0, 2.0, B)
    ((b - 0.5 * a) * w.to(d)).sum().backward()
    zr, mr, lr = (t.clone().double().requires_grad_(True) for t in (zc, muc, lvc))
    ar, br = OM.weighted(OM.log_density_plain(zr.unsqueeze(1), mr.unsqueeze(0), lr.unsqueeze(0)), N)
    ((br - 0.5 * ar) * w.double()).sum().backward()
    assert rel_err(zz.grad, zr.grad) < 1e-4 and rel_err(mm.grad, mr.grad) < 1e-4 and rel_err(ll.grad, lr.grad) < 1e-4
    # ops.py:118-122
    x = torch.randn(7, 7, generator=torch.Generator().manual_seed(3))
    dg, off = ops.on_off_diag(x.to(d))
    assert torch.equal(dg.cpu(), torch.diagonal(x)) and torch.equal(off.cpu(), x - torch.diag_embed(x))
    x1 = x[:1].contiguous()
    dg, off = ops.on_off_diag(x1.to(d))
    assert torch.equal(dg.cpu(), torch.diagonal(x1)) and torch.equal(off.cpu(), x1 - torch.diag_embed(x1))


def test_tc_c4_eight_shards_against_golden(HF):
    """BASELINE configs[3] (c4): global batch 512 as 8 data-parallel shards of 64 rows.  Every shard evaluates its
    rows with the all-gathered means (``mu_all``) and its global ``row_offset`` -- what each rank of the 8-GPU run
    computes (ops.py:52-115 on the full batch) -- forward and backward against the values the unmodified reference
    produced on the whole [512,128] batch (tests/golden/ops_c4.npz): per-row TC, the (beta-1)TC+KL loss at beta=512
    and 0.5 with d/dz, d/dmu (sum of the shards' contributions = the reduce-scatter) and d/dlogvar, and the
    per-row-weighted form.  Tolerance 1e-4 relative (north_star)."""
    import ops
    G = np.load(os.path.join(GOLDEN, "ops_c4.npz"))
    B, D, N = (int(v) for v in G["d_BDN"])
    assert (B, D, N) == (512, 128, 10000)
    R, Bl = 8, 64
    d = dev()
    z, mu, lv = (torch.from_numpy(G[f"d_{k}"]).to(d) for k in ("z", "mu", "logvar"))
    T = lambda k: torch.from_numpy(G[f"d_{k}"])  # noqa: E731

    def sharded_tc(zz, mm, ll):
        return torch.cat([ops.total_correlation(zz[r * Bl:(r + 1) * Bl], mm[r * Bl:(r + 1) * Bl], ll[r * Bl:(r + 1) * Bl],
                                                N, "none", mu_all=mm, row_offset=r * Bl) for r in range(R)])

    assert rel_err(sharded_tc(z, mu, lv), T("tc_none")) < 1e-4
    assert rel_err(sharded_tc(z, mu, lv).mean(), T("tc_mean")) < 1e-4
    for beta, bt in ((512.0, "512p0"), (0.5, "0p5")):
        zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, lv))
        loss = (beta - 1.0) * sharded_tc(zz, mm, ll).mean() + ops.kl_divergence(ll, mm, "mean")
        loss.backward()
        assert rel_err(loss, T(f"tckl_b{bt}")) < 1e-4
        assert rel_err(zz.grad, T(f"tckl_b{bt}_dz")) < 1e-4
        assert rel_err(mm.grad, T(f"tckl_b{bt}_dmu")) < 1e-4
        assert rel_err(ll.grad, T(f"tckl_b{bt}_dlogvar")) < 1e-4
    zz, mm, ll = (t.clone().requires_grad_(True) for t in (z, mu, lv))
    (T("tcw_w").to(d) * sharded_tc(zz, mm, ll)).sum().backward()
    assert rel_err(zz.grad, T("tcw_dz")) < 1e-4
    assert rel_err(mm.grad, T("tcw_dmu")) < 1e-4
    assert rel_err(ll.grad, T("tcw_dlogvar")) < 1e-4
    # the sampler components a rank reports: stratified and weighted, per shard
    for weighted, key in ((False, "mss"), (True, "mws")):
        flags = HF.abi.TC_VAR_FROM_ROW | HF.abi.TC_EPS_DENSITY | (HF.abi.TC_WEIGHTED if weighted else 0)
        pm, lq = [], []
        for r in range(R):
            a, b, _ = HF.tc_components(z[r * Bl:(r + 1) * Bl], mu, lv[r * Bl:(r + 1) * Bl], N, r * Bl, flags)
            pm.append(a), lq.append(b)
        assert rel_err(torch.cat(pm), T(f"{key}_prodm")) < 1e-4 and rel_err(torch.cat(lq), T(f"{key}_logqz")) < 1e-4


def test_tc_sharded_rows_equal_full_batch(HF):
    """Rows [r*Bl, (r+1)*Bl) with row_offset and the full mu reproduce the full-batch estimator
    (what each data-parallel rank computes), including the special logW entries."""
    import ops
    g = torch.Generator().manual_seed(4)
    B, D, N, R = 32, 24, 5000, 4
    d = dev()
    mu = torch.randn(B, D, generator=g).to(d)
    lv = (-3 + 2 * torch.randn(B, D, generator=g)).to(d)
    z = (mu + torch.randn(B, D, generator=g).to(d) * (0.5 * lv).exp())
    full = ops.total_correlation(z, mu, lv, N, "none")
    Bl = B // R
    parts = [ops.total_correlation(z[r * Bl:(r + 1) * Bl], mu[r * Bl:(r + 1) * Bl], lv[r * Bl:(r + 1) * Bl], N, "none",
                                   mu_all=mu, row_offset=r * Bl) for r in range(R)]
    assert rel_err(torch.cat(parts), full) < 1e-6


def test_reconstruction(HF):
    import ops
    G = np.load(os.path.join(GOLDEN, "ops.npz"))
    d = dev()
    x, xr = torch.from_numpy(G["rec_x"]).to(d), torch.from_numpy(G["rec_xr"]).to(d)
    for lt in ("mse", "l1", "bce"):
        for red in ("sum", "mean", "none"):
            assert rel_err(ops.reconstruction_loss(x, xr, lt, red), torch.from_numpy(G[f"rec_{lt}_{red}"])) < 1e-5
        xg = xr.clone().requires_grad_(True)
        (torch.from_numpy(G[f"rec_{lt}_w"]).to(d) * ops.reconstruction_loss(x, xg, lt, "none")).sum().backward()
        assert rel_err(xg.grad, torch.from_numpy(G[f"rec_{lt}_dxr"])) < 1e-5
    # known answers of the reference's tests/test_ops.py:10-43
    x0, x1 = torch.zeros(3, device=d), torch.tensor([1.0, 2.0, 4.0], device=d)
    assert ops.reconstruction_loss(x0, x1, "mse", "sum").item() == 21
    assert ops.reconstruction_loss(x0, x1, "mse", "mean").item() == 7
    assert ops.reconstruction_loss(x0, x1, "mse", "none").tolist() == [1, 4, 16]
    assert ops.reconstruction_loss(x0, x1, "l1", "sum").item() == 7
    assert ops.reconstruction_loss(x0, x1, "l1", "mean").item() == pytest.approx(7 / 3)
    with pytest.raises(NotImplementedError):
        ops.reconstruction_loss(x0, x1, "huber", "sum")
    with pytest.raises(NotImplementedError):
        ops.reconstruction_loss(x0, x1, "mse", "avg")
    # large rows exercise the split reduction
    g = torch.Generator().manual_seed(8)
    a, b = torch.rand(8, 3, 64, 64, generator=g), torch.rand(8, 3, 64, 64, generator=g)
    ref = ((a.double() - b.double()) ** 2).reshape(8, -1).sum(1)
    assert rel_err(ops.reconstruction_loss(a.to(d), b.to(d), "mse", "none"), ref) < 1e-6


def test_fused_loss_heads_equal_the_reference_expression_trees(HF):
    """Round 3's one-launch loss heads against the expression trees they replace, built in fp64 from torch primitives the
    way the reference's hooks / solver build them (solvers/vae.py:63-91, solvers/tc.py:69-89, solvers/intro.py:100-170):
    beta * kl_divergence, beta * reconstruction_loss, (beta-1) TC + KL, mean exp(-2 s (rec + kl)), the scalar linear
    combinations -- values and gradients, 1e-5 relative (TC term: the 1e-4 of the estimator's own golden test)."""
    import ops
    from oracle import latent_math as OM
    d = dev()
    g = torch.Generator().manual_seed(21)
    B, D = 16, 24
    mu, lv = torch.randn(B, D, generator=g), 0.7 * torch.randn(B, D, generator=g) - 0.5
    z = mu + torch.exp(0.5 * lv) * torch.randn(B, D, generator=g)
    w = torch.linspace(-1.0, 2.0, B)

    def kl_ref(l, m, red):
        rows = -0.5 * (1 + l - m.pow(2) - l.exp()).sum(1)
        return {"sum": rows.sum(), "mean": rows.mean()}.get(red, rows)

    for red in ("sum", "mean", "none"):
        for scale in (1.0, 0.37, 512.0):
            mm, ll = (t.clone().to(d).requires_grad_(True) for t in (mu, lv))
            out = ops.kl_divergence(ll, mm, red, scale=scale)
            mr, lr = (t.clone().double().requires_grad_(True) for t in (mu, lv))
            ref = scale * kl_ref(lr, mr, red)
            ((out * w.to(d)).sum() if red == "none" else out).backward()
            ((ref * w.double()).sum() if red == "none" else ref).backward()
            assert rel_err(out, ref) < 1e-5 and rel_err(mm.grad, mr.grad) < 1e-5 and rel_err(ll.grad, lr.grad) < 1e-5
    x, xr = torch.rand(B, 3, 8, 8, generator=g), torch.rand(B, 3, 8, 8, generator=g) * 0.98 + 0.01
    for lt in ("mse", "l1", "bce"):
        for red in ("sum", "mean", "none"):
            xg = xr.clone().to(d).requires_grad_(True)
            out = ops.reconstruction_loss(x.to(d), xg, lt, red, scale=0.25)
            rg = xr.clone().double().requires_grad_(True)
            a, b = x.double().reshape(B, -1), rg.reshape(B, -1)
            rows = {"mse": lambda: ((a - b) ** 2).sum(1), "l1": lambda: (a - b).abs().sum(1),
                    "bce": lambda: -(a * b.log() + (1 - a) * (1 - b).log()).sum(1)}[lt]()
            ref = 0.25 * {"sum": rows.sum(), "mean": rows.mean()}.get(red, rows)
            ((out * w.to(d)).sum() if red == "none" else out).backward()
            ((ref * w.double()).sum() if red == "none" else ref).backward()
            assert rel_err(out, ref) < 1e-5 and rel_err(xg.grad, rg.grad) < 1e-5, (lt, red)
    # (beta - 1) TC + KL, the TC solvers' hook, both reductions, also sharded over the rows (mu_all + row_offset)
    N = 5000
    for beta in (512.0, 0.5):
        for red in ("mean", "none"):
            zz, mm, ll = (t.clone().to(d).requires_grad_(True) for t in (z, mu, lv))
            out = ops.tc_kl_loss(zz, mm, ll, N, beta, red)
            zr, mr, lr = (t.clone().double().requires_grad_(True) for t in (z, mu, lv))
            a, b = OM.stratified(OM.log_density_clamped_var(zr.unsqueeze(1), mr.unsqueeze(0), lr.unsqueeze(1)), N)
            tc = b - a
            ref = (beta - 1.0) * (tc.mean() if red == "mean" else tc) + kl_ref(lr, mr, red)
            ((out * w.to(d)).sum() if red == "none" else out).backward()
            ((ref * w.double()).sum() if red == "none" else ref).backward()
            assert rel_err(out, ref) < 1e-4, (beta, red)
            assert rel_err(zz.grad, zr.grad) < 1e-4 and rel_err(mm.grad, mr.grad) < 1e-4 and rel_err(ll.grad, lr.grad) < 1e-4
            if red == "none":
                parts = [ops.tc_kl_loss(z[r * 8:(r + 1) * 8].to(d), mu[r * 8:(r + 1) * 8].to(d), lv[r * 8:(r + 1) * 8].to(d), N,
                                        beta, red, mu_all=mu.to(d), row_offset=r * 8) for r in range(2)]
                assert rel_err(torch.cat(parts), out) < 1e-6
    # solvers/intro.py:102-103: exp(-2 s (rec + beta kl)).mean()
    a, b = 30.0 * torch.rand(B, generator=g), 5.0 * torch.rand(B, generator=g)
    for c in (-2.0 / 12288, -0.05):
        ag, bg = (t.clone().to(d).requires_grad_(True) for t in (a, b))
        out = HF.ExpElboFn.apply(ag, bg, c)
        ar, br = (t.clone().double().requires_grad_(True) for t in (a, b))
        ref = torch.exp(c * (ar + br)).mean()
        (3.0 * out).backward()
        (3.0 * ref).backward()
        assert rel_err(out, ref) < 1e-6 and rel_err(ag.grad, ar.grad) < 1e-5 and rel_err(bg.grad, br.grad) < 1e-5
    # the solver's scalar algebra: sum_k w_k t_k, gradient w_k * g for every term
    terms = [torch.randn((), generator=g) for _ in range(5)]
    wts = [0.5, -1.25, 3.0, 1.0 / 12288, 0.0]
    tg = [t.clone().to(d).requires_grad_(True) for t in terms]
    out = HF.LinCombFn.apply(wts, *tg)
    (2.0 * out).backward()
    assert rel_err(out, sum(wk * t.double() for wk, t in zip(wts, terms))) < 1e-6
    assert [float(t.grad) for t in tg] == pytest.approx([2.0 * wk for wk in wts], rel=1e-6, abs=1e-12)
    with pytest.raises(Exception):
        HF.LinCombFn.apply([1.0], torch.zeros(2, device=d))


def test_ops_shapes_like_reference_tests():
    """Shape checks of the reference's tests/test_ops.py:48-66 on the HIP path."""
    import ops
    d = dev()
    mu = torch.zeros(3, device=d).view(1, 3)
    logvar = torch.tensor([[1.0, 2.0, 4.0]], device=d)
    assert ops.reparameterize(mu, logvar).shape == mu.shape
    mu2 = torch.zeros(2, 2, device=d)
    lv2 = torch.tensor([[1.0, 2.0], [4.0, 8.0]], device=d)
    assert ops.kl_divergence(mu2, lv2, reduce="sum").dim() == 0
    assert ops.kl_divergence(mu2, lv2, reduce="none").dim() == 1


def test_adam_and_clip_match_torch():
    from hipvae.flat import FlatGroup, clip_grad_norm
    g = torch.Generator().manual_seed(6)
    shapes = [(7, 3, 3, 3), (13,), (5, 11), (1,)]
    ps_ref = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    ps = [torch.nn.Parameter(p.detach().clone().to(dev())) for p in ps_ref]
    opt = torch.optim.Adam(ps_ref, lr=2e-4)
    grp = FlatGroup(ps)
    for step in range(3):
        grads = [torch.randn(*s, generator=g) * (50.0 if step == 1 else 0.01) for s in shapes]
        for p, pr, gr in zip(ps, ps_ref, grads):
            pr.grad = gr.clone()
            p.grad.copy_(gr.to(dev()))
        n_ref = torch.nn.utils.clip_grad_norm_(ps_ref, 100.0)
        n = clip_grad_norm([grp], 100.0)
        assert abs(float(n) - float(n_ref)) <= 1e-5 * float(n_ref)
        for p, pr in zip(ps, ps_ref):
            assert rel_err(p.grad, pr.grad) < 1e-6
        opt.step()
        grp.adam_step(2e-4)
        for p, pr in zip(ps, ps_ref):
            assert float((p.detach().cpu() - pr.detach()).abs().max()) < 2e-9 + 1e-6 * 2e-4


def test_errors_are_loud():
    from hipvae import abi
    with pytest.raises(abi.HipExtensionError):
        abi.ptr(torch.zeros(3))
    with pytest.raises(RuntimeError):
        abi.call("itcv_conv2d_pack_weight", None, None, 1, 1, 7, 0, None)
    assert "itcv_conv2d_pack_weight" in abi.last_error()


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("planes", [0, 2])
@pytest.mark.parametrize("dims", [(2, 6, 16, 8, 8), (2, 8, 32, 32, 32), (3, 4, 64, 16, 16)])
def test_bn_groups_equal_separate_passes(HF, pool, planes, dims):
    """BatchNorm groups (models.bn_groups): one call on G stacked passes == G separate calls -- outputs, the planes
    handed to the consumer conv, running buffers (advanced once per pass, in order), and every gradient, bit for bit
    (the same kernels run on the same sub-batches)."""
    G, Bg, C, H, W = dims      # one-block-per-channel statistics (first) and sliced statistics (the other two)
    g = torch.Generator().manual_seed(31)
    d = dev()
    x = (torch.randn(G * Bg, C, H, W, generator=g) * 1.5 + 0.3).to(d)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(d), torch.randn(C, generator=g).to(d)
    oshape = (G * Bg, C, H // 2, W // 2) if pool else (G * Bg, C, H, W)
    dy = torch.randn(*oshape, generator=g).to(d)

    def run(groups):
        rm, rv, nbt = torch.zeros(C, device=d), torch.ones(C, device=d), torch.zeros((), dtype=torch.long, device=d)
        xs, ga, be = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        if groups > 1:
            y = HF.BnActFn.apply(xs, ga, be, None, rm, rv, nbt, 1e-4, 0.1, 0.2, pool, True, None, planes, planes, True, True, groups)
            pl = HF._tagged_planes(y, planes) if planes else None
        else:
            ys, pls = [], []
            for k in range(G):
                yk = HF.BnActFn.apply(xs[k * Bg:(k + 1) * Bg], ga, be, None, rm, rv, nbt, 1e-4, 0.1, 0.2, pool, True, None,
                                      planes, planes, True, True, 1)
                ys.append(yk)
                pls.append(HF._tagged_planes(yk, planes) if planes else None)
            y, pl = torch.cat(ys), pls
        y.backward(dy)
        return y.detach(), pl, rm, rv, int(nbt), xs.grad, ga.grad, be.grad

    a, b = run(G), run(1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[4] == b[4] == G
    assert torch.equal(a[5], b[5])
    assert rel_err(a[6], b[6]) < 1e-6 and rel_err(a[7], b[7]) < 1e-6     # two fp32 additions in a different grouping
    if planes:
        # planes of the batched tensor: [plane][image][c/8][hw]; a pass's planes are a sub-range of every plane
        n_img = C // 8 * oshape[2] * oshape[3] * 4           # int32 words per image per plane
        whole = a[1].view(planes, G * Bg, n_img)
        for k in range(G):
            assert torch.equal(whole[:, k * Bg:(k + 1) * Bg], b[1][k].view(planes, Bg, n_img))
    # statistics really are per pass: a single-group call on the stacked batch differs
    rm, rv, nbt = torch.zeros(C, device=d), torch.ones(C, device=d), torch.zeros((), dtype=torch.long, device=d)
    y1 = HF.BnActFn.apply(x, gamma, beta, None, rm, rv, nbt, 1e-4, 0.1, 0.2, pool, True, None)
    assert not torch.equal(y1, a[0])
