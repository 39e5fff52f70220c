"""GPU tests of the pre-split "planes" pipeline (include/itcv_hip.h: itcv_split_planes, itcv_conv2d_fwd_bf16p,
itcv_conv2d_wgrad_bf16p, itcv_bn_train_fwd / _bwd with planes): the operand format of the split-bf16 hot GEMMs.

* planes reconstruct the tensor (2^-16 relative with two bf16 planes, fp32-exact class with three);
* the LDS-DMA kernels on planes give BIT-IDENTICAL results to the gather kernels on the fp32 tensor where the
  K decomposition is the same (no split-K), and agree to rounding where it differs;
* the transposing-read weight gradient against an fp64 reference (5e-5 of the result scale in bf16x3);
* BatchNorm passes that emit planes (and fold their statistics in the apply launch) against an fp64 reference,
  with the planes equal to a split of the fp32 output.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def HF():
    from hipvae import functional
    return functional


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def unpack_planes(xp, shape, ns):
    """planes [ns][B][C/8][H*W] x 8 bf16 -> fp32 [B,C,H,W] (sum of the planes)."""
    B, C, H, W = shape
    raw = xp.view(torch.int16).view(ns, B, C // 8, H * W, 8)                 # bf16 bit patterns
    vals = (raw.to(torch.int32) << 16).view(torch.float32)                  # bf16 -> fp32 is a 16-bit shift
    return vals.sum(0).permute(0, 1, 3, 2).reshape(B, C, H, W)


@pytest.mark.parametrize("ns,tol", [(2, 2.0 ** -15), (3, 2.0 ** -22)])
def test_split_planes_reconstructs(HF, ns, tol):
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(3, 16, 8, 12, generator=g) * torch.logspace(-3, 3, 16).view(1, 16, 1, 1)).to(dev())
    xp = HF.split_planes(x, ns)
    back = unpack_planes(xp, x.shape, ns)
    assert float(((back - x).abs() / x.abs().clamp_min(1e-30)).max()) < tol


PLANES_CASES = [  # B, Ci, H, W, Co, up2 -- band kernel (W in 8..64), 128-pixel-tile kernel (W = 4), 64- and 128-row tiles
    (2, 64, 16, 16, 64, False), (2, 128, 8, 8, 160, False), (3, 32, 32, 32, 48, False), (2, 64, 64, 64, 64, False),
    (4, 128, 4, 4, 256, False), (2, 64, 16, 16, 128, True), (2, 96, 32, 32, 64, True),
    # 128- and 256-wide images: 128-pixel tiles of the persistent band kernel (one row / half a row per tile)
    (2, 64, 8, 128, 64, False), (1, 64, 8, 256, 64, False), (2, 32, 16, 128, 160, False), (1, 64, 8, 256, 64, True),
    (3, 64, 4, 128, 128, True), (5, 64, 128, 128, 64, False),
]


@pytest.mark.parametrize("case", PLANES_CASES)
@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6"])
def test_planes_conv_equals_gather_conv(HF, mode, case):
    """Forward and data-gradient on planes vs the kernels that gather + split the fp32 tensor themselves."""
    B, Ci, H, W, Co, up2 = case
    ns = 2 if mode == "bf16x3" else 3
    g = torch.Generator().manual_seed(sum(case[:5]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g).to(dev())
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(dev())
    dy = torch.randn(B, Co, H, W, generator=g).to(dev())
    with HF.conv_math_scope(mode):
        ref = HF.conv_apply(x, w, w, 0, None, B, Ci, H, W, Co, 3, up2)
        got = HF.conv_apply_planes(HF.split_planes(x, ns), w, w, 0, None, B, Ci, H, W, Co, 3, up2, ns)
        pairs = [(got, ref)]
        if HF.lib.itcv_conv2d_bf16s_supported(Co, Ci, 3):   # data-gradient: the channel roles swap
            refd = HF.conv_apply(dy, w, w, 1, None, B, Co, H, W, Ci, 3, False)
            gotd = HF.conv_apply_planes(HF.split_planes(dy, ns), w, w, 1, None, B, Co, H, W, Ci, 3, False, ns)
            pairs.append((gotd, refd))
    for a, b in pairs:
        assert torch.equal(a, b) or rel_err(a, b) < 2e-6   # equal unless the two kernels split K differently


# Launches with MORE TILES THAN CUs at W <= 64: the persistent band kernel (conv_fwd_bf16p3_kernel: tile walk, next-tile
# band prefetch, weight-ring wrap, stores draining under the next tile) in every instantiation the c2 step uses
# (LOG2W 6/5/4/3 x 64/128-row tiles x up2) -- the kernel bench.py's roofline is quoted on.
PERSIST_CASES = [  # B, Ci, H, W, Co, up2
    (20, 64, 64, 64, 64, False), (20, 64, 64, 64, 64, True),            # LOG2W 6, BM 64: 320 tiles
    (72, 128, 32, 32, 128, False), (36, 64, 32, 32, 128, False),        # LOG2W 5, BM 128 (288 tiles) / BM 64 (2 x 144)
    (72, 64, 32, 32, 64, True),                                         # LOG2W 5, BM 64, up2
    (136, 64, 16, 16, 256, True), (272, 64, 16, 16, 64, False),         # LOG2W 4, BM 128 up2 (2 x 136) / BM 64
    (520, 128, 8, 8, 128, False), (1040, 64, 8, 8, 256, False),         # LOG2W 3, BM 64 (2 x 130) / BM 128 (2 x 260)
]


@pytest.mark.parametrize("case", PERSIST_CASES)
def test_persistent_band_kernel_vs_one_tile_kernel_and_fp64(HF, case):
    """Forward and data-gradient (bf16x3, planes): the persistent kernel with its default 256 blocks, with 5 blocks (long
    tile walks, ragged tile lists) and the one-tile-per-block kernel (band_persist_blocks = 0) are BIT-IDENTICAL, and
    within 5e-5 of the result scale of an fp64 convolution (checked on the first / last two images; the reference the
    kernel replaces: ATen conv forward / data-gradient, /root/reference/models.py:28-47)."""
    B, Ci, H, W, Co, up2 = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    dy = torch.randn(B, Co, H, W, generator=g)
    xd, wd, dyd = x.to(dev()), w.to(dev()), dy.to(dev())
    sel = [0, 1, B - 2, B - 1]

    def run():
        with HF.conv_math_scope("bf16x3"):
            y = HF.conv_apply_planes(HF.split_planes(xd, 2), wd, wd, 0, None, B, Ci, H, W, Co, 3, up2, 2)
            dx = HF.conv_apply_planes(HF.split_planes(dyd, 2), wd, wd, 1, None, B, Co, H, W, Ci, 3, False, 2)
        return y, dx

    assert HF.get_option("band_persist_blocks") == 256
    y, dx = run()
    for blocks in (0, 5):
        with HF.option_scope("band_persist_blocks", blocks):
            y2, dx2 = run()
        assert torch.equal(y, y2) and torch.equal(dx, dx2), blocks
    xin = F.interpolate(x[sel].double(), scale_factor=2, mode="nearest") if up2 else x[sel].double()
    assert rel_err(y[sel], F.conv2d(xin, w.double(), padding=1)) < 5e-5
    assert rel_err(dx[sel], F.conv_transpose2d(dy[sel].double(), w.double(), padding=1)) < 5e-5
    # the other inner product (v_mfma_f32_32x32x16_bf16): same values to rounding
    with HF.option_scope("band_m16", 0):
        y3, dx3 = run()
    assert rel_err(y3, y) < 2e-6 and rel_err(dx3, dx) < 2e-6 and not torch.equal(y3, y)


@pytest.mark.parametrize("case", [(4, 128, 4, 4, 256, False), (32, 512, 4, 4, 256, False), (16, 256, 4, 4, 544, True)])
@pytest.mark.parametrize("math", ["bf16x3", "f16x3"])
def test_planes_kernel_four_and_eight_mfma_waves_bit_identical(HF, case, math):
    """The 128 x 128 tile of the 128-pixel planes kernel (W = 4 layers, split-K included) with eight MFMA waves of 64 x 32
    (default) and with four of 64 x 64: the same K order, bit-identical forward and data-gradient."""
    B, Ci, H, W, Co, up2 = case
    ns = {"bf16x3": 2, "f16x3": 4}[math]
    g = torch.Generator().manual_seed(sum(case[:5]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g).to(dev())
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(dev())
    dy = torch.randn(B, Co, H, W, generator=g).to(dev())
    out = {}
    with HF.conv_math_scope(math):
        for nw in (8, 4):
            with HF.option_scope("planes_mfma_waves", nw):
                y = HF.conv_apply_planes(HF.split_planes(x, ns), w, w, 0, None, B, Ci, H, W, Co, 3, up2, ns)
                dx = HF.conv_apply_planes(HF.split_planes(dy, ns, gradient=True), w, w, 1, None, B, Co, H, W, Ci, 3, False, ns)
            out[nw] = (y, dx)
    assert torch.equal(out[8][0], out[4][0]) and torch.equal(out[8][1], out[4][1])


def test_option_api_validates(HF):
    """itcv_set_option: unknown names and out-of-range values are refused; nothing is read from the environment."""
    from hipvae import abi
    for name, value in (("band_persist_blocks", -1), ("band_persist_blocks", 4096), ("band_m16", 2), ("wgrad_m16", 2),
                        ("planes_mfma_waves", 6), ("no_such_option", 1)):
        with pytest.raises(RuntimeError):
            HF.set_option(name, value)
    assert HF.get_option("no_such_option") == -1
    assert HF.get_option("band_m16") == 1 and HF.get_option("band_persist_blocks") == 256
    assert HF.get_option("wgrad_m16") == 1 and HF.get_option("planes_mfma_waves") == 8


def test_batched_bn_act_conv_chain_vs_fp64(HF):
    """The c2 step's heaviest chain at its batched size: 2 x 64 images of 64 channels at 64x64 -- BatchNorm (two groups,
    train mode) + LeakyReLU emitting PLANES ONLY, then the 64 -> 64 3x3 conv on the persistent band kernel (2048 tiles) --
    against fp64 BatchNorm over the whole batch and an fp64 conv on the first / last images (models.py:37-47)."""
    B, C, S = 128, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, S, S, generator=g) * 1.3 + 0.2
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    d = dev()
    rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    with HF.conv_math_scope("bf16x3"):
        a = HF.BnActFn.apply(x.to(d), gamma.to(d), beta.to(d), None, rm, rv, nbt, 1e-4, 0.1, 0.2, False, True, None, 2, 0,
                             False, True, 2)
        assert HF._tagged_planes(a, 2) is not None and not a._itcv_planes[4]      # planes only: the fp32 tensor is not written
        y = HF.Conv2dFn.apply(a, w.to(d), None, False)
    assert int(nbt) == 2
    sel = [0, 1, 63, 64, 126, 127]
    ref = []
    for grp in (slice(0, 64), slice(64, 128)):
        xg = x[grp].double()
        ref.append(F.leaky_relu(F.batch_norm(xg, None, None, gamma.double(), beta.double(), True, 0.1, 1e-4), 0.2))
    act = torch.cat(ref)
    assert rel_err(y[sel], F.conv2d(act[sel], w.double(), padding=1)) < 5e-5


WGRAD_CASES_F16 = [(2, 64, 16, 16, 64, False), (4, 128, 4, 4, 256, False), (2, 32, 32, 32, 48, False), (1, 64, 64, 64, 64, False),
                   (2, 128, 16, 16, 64, True), (2, 512, 8, 8, 256, False), (1, 64, 8, 128, 64, False), (3, 64, 64, 64, 64, True)]


# ---- fp16 planes ("f16x3": hi / lo fp16 planes of S * x, 3 products) ---------------------------------------------------
def unpack_f16_planes(xp, shape):
    """fp16 planes [2][B][C/8][H*W] x 8 fp16 + scale record -> (fp32 [B,C,H,W] = (hi + lo) / S, S)."""
    B, C, H, W = shape
    n = 2 * B * (C // 8) * H * W * 4                                        # int32 words of the two planes
    rec = xp[n:n + 4].view(torch.float32)
    scale, inv = float(rec[0]), float(rec[1])
    assert scale > 0 and scale * inv == 1.0 and float(torch.tensor(scale).log2()) % 1 == 0     # an exact power of two
    vals = xp[:n].view(torch.float16).view(2, B, C // 8, H * W, 8).float()
    return (vals.sum(0) * inv).permute(0, 1, 3, 2).reshape(B, C, H, W), scale


@pytest.mark.parametrize("magnitude", [1.0, 3e-9, 2e5])
def test_f16_planes_reconstruct(HF, magnitude):
    """Two fp16 planes carry 22 significand bits of S*x: a gradient-like tensor of ANY magnitude (1e-9 .. 1e5, heavy
    tailed) comes back to 2^-20.5 of each element (elements >= 2^-18 of the largest) and 2^-38 of the largest below that;
    the scale is the power of two that puts the largest element in [2^14, 2^15).  Activations (scale 1) alike."""
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(3, 16, 8, 12, generator=g) * torch.logspace(-2, 0, 16).view(1, 16, 1, 1) * magnitude).to(dev())
    xp = HF.split_planes(x, HF.F16X2, gradient=True)
    back, scale = unpack_f16_planes(xp, x.shape)
    top = float(x.abs().max())
    assert 2.0 ** 14 <= top * scale < 2.0 ** 15
    err = (back - x).abs()
    big = x.abs() >= top * 2.0 ** -18
    assert float((err[big] / x.abs()[big]).max()) < 2.0 ** -20.5
    if (~big).any():
        assert float(err[~big].max()) < top * 2.0 ** -38
    if magnitude == 1.0:
        back1, s1 = unpack_f16_planes(HF.split_planes(x, HF.F16X2), x.shape)
        assert s1 == 1.0 and float(((back1 - x).abs() / x.abs().clamp_min(2.0 ** -3)).max()) < 2.0 ** -20.5


@pytest.mark.parametrize("case", PLANES_CASES[:7] + PLANES_CASES[7:9] + PERSIST_CASES[:2])
def test_f16_planes_conv_vs_fp64(HF, case):
    """Forward and data-gradient on fp16 planes (band, persistent band, 128-pixel kernels): within 1.5e-6 of the result
    scale of an fp64 convolution -- the level of the exact-fp32 MFMA kernel (fp32 accumulation), 30x below bf16x3 -- with
    the data-gradient's input at gradient-like magnitude (1e-7)."""
    B, Ci, H, W, Co, up2 = case
    B = min(B, 6)
    g = torch.Generator().manual_seed(sum(case[:5]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    dy = torch.randn(B, Co, H, W, generator=g) * 1e-7
    xd, wd, dyd = x.to(dev()), w.to(dev()), dy.to(dev())
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
    ref = F.conv2d(xin, w.double(), padding=1)
    refd = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    with HF.conv_math_scope("f16x3"):
        y = HF.conv_apply_planes(HF.split_planes(xd, 4), wd, wd, 0, None, B, Ci, H, W, Co, 3, up2, 4)
        dx = None
        if HF.lib.itcv_conv2d_bf16s_supported(Co, Ci, 3):   # data-gradient: the channel roles swap
            dx = HF.conv_apply_planes(HF.split_planes(dyd, 4, gradient=True), wd, wd, 1, None, B, Co, H, W, Ci, 3, False, 4)
    with HF.conv_math_scope("fp32"):
        y32 = HF.conv_apply(xd, wd, wd, 0, None, B, Ci, H, W, Co, 3, up2)
    e16, e32 = rel_err(y, ref), rel_err(y32, ref)
    assert e16 < 1.5e-6 and e16 < 3 * e32 + 2e-7, (e16, e32)
    assert dx is None or rel_err(dx, refd) < 1.5e-6


@pytest.mark.parametrize("case", WGRAD_CASES_F16)
def test_f16_planes_weight_gradient(HF, case):
    """Weight gradient from fp16 planes (x at O(1), dy at 1e-8): 1.5e-6 of the result scale vs fp64, bitwise repeatable."""
    B, Ci, H, W, Co, up2 = case
    g = torch.Generator().manual_seed(7 + sum(case[:5]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g) * 1e-8
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
    ref = torch.nn.grad.conv2d_weight(xin, (Co, Ci, 3, 3), dy.double(), padding=1)
    xp, dyp = HF.split_planes(x.to(dev()), 4), HF.split_planes(dy.to(dev()), 4, gradient=True)
    dw = HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, ns=4)
    assert rel_err(dw, ref) < 1.5e-6
    assert torch.equal(dw, HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, ns=4))
    for m16 in (0, 1):
        with HF.option_scope("wgrad_m16", m16):
            assert rel_err(HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, ns=4), ref) < 1.5e-6, m16


@pytest.mark.parametrize("shape,pool,groups", [((4, 64, 32, 32), False, 1), ((4, 64, 32, 32), True, 2), ((8, 128, 16, 16), False, 2),
                                               ((16, 512, 4, 4), False, 2), ((2, 16, 8, 8), False, 1)])
@pytest.mark.parametrize("gscale", [1.0, 1e-9])
def test_batchnorm_emits_f16_planes(HF, shape, pool, groups, gscale):
    """BnActFn with fp16 planes: the forward planes reconstruct y (scale 1); the backward planes reconstruct dx with the
    scale the apply pass derives from the partial pass's maxima -- a power of two, with max|dx| * S below 2^15 (never
    overflowing) and above 2^2 (bound loose by at most 2^13: typical values keep all 22 bits) -- in every launch form
    (sliced / one block per channel, one or two BatchNorm groups), at O(1) and at 1e-9 gradient magnitude."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B + C + H + groups)
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    oshape = (B, C, H // 2, W // 2) if pool else shape
    dy = torch.randn(*oshape, generator=g) * gscale
    d = dev()
    rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    xd = x.to(d).requires_grad_(True)
    seen = []

    class Tap(torch.autograd.Function):      # hands the gradient OBJECT BnActFn returns (with its planes tag) to the test
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            seen.append(g)
            return g

    y = HF.BnActFn.apply(Tap.apply(xd), gamma.to(d), beta.to(d), None, rm, rv, nbt, 1e-4, 0.1, 0.2, pool, True, None, 4, 4, True, True,
                         groups)
    yp = HF._tagged_planes(y, 4)
    assert yp is not None
    back, s = unpack_f16_planes(yp, y.shape)
    assert s == 1.0 and float((back - y.detach()).abs().max()) < 2.0 ** -21 * float(y.detach().abs().max())
    y.backward(dy.to(d))
    dx = seen[0]
    dxp = HF._tagged_planes(dx, 4)
    assert dxp is not None and torch.equal(dx, xd.grad)
    backd, sd = unpack_f16_planes(dxp, dx.shape)
    top = float(dx.abs().max())
    assert 2.0 ** 2 < top * sd < 2.0 ** 15, (top * sd, sd)
    assert float((backd - dx).abs().max()) < top * 2.0 ** -21
    # same values as the bf16-planes form computes for dx (the fp32 tensor does not depend on the plane format)
    xd2 = x.to(d).requires_grad_(True)
    y2 = HF.BnActFn.apply(xd2, gamma.to(d), beta.to(d), None, rm.clone(), rv.clone(), nbt.clone(), 1e-4, 0.1, 0.2, pool, True,
                          None, 2, 2, True, True, groups)
    y2.backward(dy.to(d))
    assert torch.equal(y2, y) and torch.equal(xd2.grad, dx)


def test_f16_small_layer_kernels_vs_fp64(HF):
    """The 5x5 stem / predict layers in fp16 planes form: 3 -> 64 forward and data-gradient (in-register split, gradient
    input at 1e-8), 64 -> 3 on planes, and both 5x5 weight gradients -- 2e-6 of the result scale vs fp64."""
    g = torch.Generator().manual_seed(9)
    B, S = 3, 64
    d = dev()
    img = torch.rand(B, 3, S, S, generator=g)
    w_stem = torch.randn(64, 3, 5, 5, generator=g) / 6.0
    w_pred = torch.randn(3, 64, 5, 5, generator=g) / 40.0
    bias = torch.randn(3, generator=g)
    act = torch.randn(B, 64, S, S, generator=g)
    gsmall = torch.randn(B, 3, S, S, generator=g) * 1e-8
    gbig = torch.randn(B, 64, S, S, generator=g) * 1e-8
    with HF.conv_math_scope("f16x3"):
        y = HF.conv_apply(img.to(d), w_stem.to(d), w_stem.to(d), 0, None, B, 3, S, S, 64, 5, False)
        assert rel_err(y, F.conv2d(img.double(), w_stem.double(), padding=2)) < 2e-6
        dxp = HF.conv_apply(gsmall.to(d), w_pred.to(d), w_pred.to(d), 1, None, B, 3, S, S, 64, 5, False)
        assert rel_err(dxp, F.conv_transpose2d(gsmall.double(), w_pred.double(), padding=2)) < 2e-6
        yp = HF.conv_apply_planes(HF.split_planes(act.to(d), 4), w_pred.to(d), w_pred.to(d), 0, bias.to(d), B, 64, S, S, 3, 5, False, 4)
        assert rel_err(yp, F.conv2d(act.double(), w_pred.double(), bias.double(), padding=2)) < 2e-6
        dxs = HF.conv_apply_planes(HF.split_planes(gbig.to(d), 4, gradient=True), w_stem.to(d), w_stem.to(d), 1, None, B, 64, S, S, 3, 5, False, 4)
        assert rel_err(dxs, F.conv_transpose2d(gbig.double(), w_stem.double(), padding=2)) < 2e-6
        dw_stem = HF.conv_wgrad5_planes(img.to(d), HF.split_planes(gbig.to(d), 4, gradient=True), B, 3, S, S, True, ns=4)
        assert rel_err(dw_stem, torch.nn.grad.conv2d_weight(img.double(), (64, 3, 5, 5), gbig.double(), padding=2)) < 2e-6
        dw_pred = HF.conv_wgrad5_planes(gsmall.to(d), HF.split_planes(act.to(d), 4), B, 3, S, S, False, ns=4)
        assert rel_err(dw_pred, torch.nn.grad.conv2d_weight(act.double(), (3, 64, 5, 5), gsmall.double(), padding=2)) < 2e-6


WGRAD_CASES = [  # B, Ci, H, W, Co, up2
    (2, 64, 16, 16, 64, False), (4, 128, 4, 4, 256, False), (2, 32, 32, 32, 48, False), (1, 64, 64, 64, 64, False),
    (2, 128, 16, 16, 64, True), (2, 512, 8, 8, 256, False), (8, 24, 8, 8, 136, False),
    # images wider than the 64-pixel band: one 64-column segment of a row per step (128x128 / 256x256 configurations)
    (1, 64, 8, 128, 64, False), (2, 32, 4, 256, 128, False), (1, 64, 16, 128, 32, True), (1, 16, 8, 256, 64, True),
    # 64 x 64 tile (both sides <= 64 channels): wave groups over the k-steps of a step, separate slabs
    (3, 64, 64, 64, 64, True), (8, 40, 8, 8, 56, False), (4, 64, 4, 4, 64, False),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_planes_weight_gradient(HF, case):
    B, Ci, H, W, Co, up2 = case
    assert HF.lib.itcv_conv2d_wgrad_bf16p_supported(B, Ci, H, W, Co, 3)
    g = torch.Generator().manual_seed(7 + sum(case[:5]))
    hs, ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, hs, ws, generator=g)
    dy = torch.randn(B, Co, H, W, generator=g)
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
    ref = torch.nn.grad.conv2d_weight(xin, (Co, Ci, 3, 3), dy.double(), padding=1)
    xp, dyp = HF.split_planes(x.to(dev()), 2), HF.split_planes(dy.to(dev()), 2)
    dw = HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2)
    assert rel_err(dw, ref) < 5e-5
    acc = HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, out=dw.clone(), accumulate=True)
    assert rel_err(acc, 2 * ref) < 5e-5
    again = HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2)
    assert torch.equal(dw, again)   # split-K slabs are reduced in a fixed order
    # both inner products in every tile form: same values to rounding, same bar vs fp64
    for m16 in (0, 1):
        with HF.option_scope("wgrad_m16", m16):
            alt = HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2)
        assert rel_err(alt, ref) < 5e-5 and rel_err(alt, dw) < 2e-6, m16
    assert HF.get_option("wgrad_m16") == 1


def test_deferred_weight_gradient_reduces_equal_immediate_ones(HF):
    """hipvae.functional.deferred_wgrad_reduces: the slab reduces of a whole backward pass folded by ONE launch
    (itcv_wgrad_reduce_many) are bitwise the per-call reduces -- several layers, a weight that two network passes of the
    backward both add to (its slabs chained in call order), accumulation into existing gradients, and a second, identical
    backward that reuses the persistent slab buffers and the cached device table."""
    g = torch.Generator().manual_seed(3)
    d = dev()
    layers = [(2, 64, 16, 16, 64, False), (4, 128, 4, 4, 256, False), (2, 32, 32, 32, 48, True), (2, 64, 16, 16, 64, False)]
    ops_ = []
    for B, Ci, H, W, Co, up2 in layers:
        hs, ws = (H // 2, W // 2) if up2 else (H, W)
        xp = HF.split_planes(torch.randn(B, Ci, hs, ws, generator=g).to(d), 2)
        dyp = HF.split_planes(torch.randn(B, Co, H, W, generator=g).to(d), 2)
        ops_.append((xp, dyp, B, Ci, H, W, Co, 3, up2))
    grads0 = [torch.randn(o[6], o[3], 3, 3, generator=g).to(d) for o in ops_[:3]]
    targets = [0, 1, 2, 0]                       # the fourth call adds to the first layer's gradient again

    def run(deferred):
        gr = [t.clone() for t in grads0]
        for _ in range(2):                       # two backward passes: the second one hits the pool / table caches
            if deferred:
                with HF.deferred_wgrad_reduces():
                    for o, t in zip(ops_, targets):
                        HF.conv_wgrad_planes(*o, out=gr[t], accumulate=True)
                    assert len(HF._DEFER["pending"]) == 4
                assert not HF._DEFER["pending"]
            else:
                for o, t in zip(ops_, targets):
                    HF.conv_wgrad_planes(*o, out=gr[t], accumulate=True)
        return gr

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("shape,pool", [((4, 64, 32, 32), False), ((4, 64, 32, 32), True), ((2, 16, 8, 8), False),
                                        ((8, 128, 16, 16), False)])
def test_batchnorm_emits_planes(HF, shape, pool):
    """BnActFn with planes outputs (statistics folded into the apply launch where the reduction is sliced)."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B + C + H)
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    oshape = (B, C, H // 2, W // 2) if pool else shape
    dy = torch.randn(*oshape, generator=g)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    yr = F.leaky_relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-4), 0.2)
    yr = F.avg_pool2d(yr, 2) if pool else yr
    yr.backward(dy.double())
    xd, gd, bd = (t.to(dev()).requires_grad_(True) for t in (x, gamma, beta))
    rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    nbt = torch.zeros((), dtype=torch.int64, device=dev())
    y = HF.BnActFn.apply(xd, gd, bd, None, rm, rv, nbt, 1e-4, 0.1, 0.2, pool, True, None, 2, 2, True, True)
    assert rel_err(y, yr) < 1e-5
    yp = HF._tagged_planes(y, 2)
    assert yp is not None and torch.equal(yp, HF.split_planes(y.detach(), 2))
    y.backward(dy.to(dev()))
    assert rel_err(xd.grad, xr.grad) < 2e-5 and rel_err(gd.grad, gr.grad) < 1e-5 and rel_err(bd.grad, br.grad) < 1e-5
    mean_ref = x.double().mean((0, 2, 3))
    assert rel_err(rm, 0.1 * mean_ref) < 1e-5 and int(nbt) == 1
    # planes-only form: the fp32 tensors are not written, the planes still carry the same values
    xd2 = x.to(dev()).requires_grad_(True)
    y2 = HF.BnActFn.apply(xd2, gd.detach(), bd.detach(), None, rm.clone(), rv.clone(), nbt.clone(), 1e-4, 0.1, 0.2, pool,
                          True, None, 2, 0, False, True)
    assert torch.equal(HF._tagged_planes(y2, 2), yp)


@pytest.mark.parametrize("B,S,Co,dgrad", [(3, 16, 3, False), (2, 64, 3, True), (5, 32, 2, False), (2, 8, 1, True)])
def test_small_cout_conv_on_planes(HF, B, S, Co, dgrad):
    """The 5x5, <= 3-output conv (predict layer / stem data-gradient) on the bf16 matrix cores from planes of its
    64-channel input: rows (co, dw), rolling row accumulators, dw fold -- against fp64 (bf16x3: 5e-5)."""
    assert HF.lib.itcv_conv2d_small_cout_bf16p_supported(64, Co, 5)
    g = torch.Generator().manual_seed(B * 100 + S + Co)
    x = torch.randn(B, 64, S, S, generator=g)
    w = torch.randn((64, Co, 5, 5) if dgrad else (Co, 64, 5, 5), generator=g) / 40.0
    bias = None if dgrad else torch.randn(Co, generator=g)
    wr = (w.flip(2, 3).transpose(0, 1) if dgrad else w).double()
    ref = F.conv2d(x.double(), wr, None if bias is None else bias.double(), padding=2)
    xp = HF.split_planes(x.to(dev()), 2)
    got = HF.conv_apply_planes(xp, w.to(dev()), w.to(dev()), int(dgrad), None if bias is None else bias.to(dev()), B, 64,
                               S, S, Co, 5, False, 2)
    assert rel_err(got, ref) < 5e-5


@pytest.mark.parametrize("B,H,W,Ci,dgrad", [(3, 32, 32, 3, False), (2, 64, 64, 3, True), (5, 20, 64, 2, False),
                                            (2, 7, 32, 1, True), (1, 128, 128, 3, False), (1, 40, 256, 3, True),
                                            (64, 64, 64, 3, False)])
def test_small_cin_conv_on_matrix_cores(HF, B, H, W, Ci, dgrad):
    """The 5x5, <= 3-channel -> 64-channel conv (stem layer / predict data-gradient) as bf16x3 products on the matrix
    cores: reduction (dw, ci) per filter row, ring of five input-row fragments -- against fp64 (bf16x3: 5e-5) and the
    direct fp32 kernel it replaces; ragged row blocks, image edges, bias."""
    assert HF.lib.itcv_conv2d_small_cin_bf16x3_supported(Ci, 64, 5, W)
    assert not HF.lib.itcv_conv2d_small_cin_bf16x3_supported(Ci, 64, 5, 48)
    assert not HF.lib.itcv_conv2d_small_cin_bf16x3_supported(4, 64, 5, W)
    g = torch.Generator().manual_seed(B * 100 + H + W + Ci)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn((Ci, 64, 5, 5) if dgrad else (64, Ci, 5, 5), generator=g) / 6.0
    bias = None if dgrad else torch.randn(64, generator=g)
    wr = (w.flip(2, 3).transpose(0, 1) if dgrad else w).double()
    ref = F.conv2d(x.double(), wr, None if bias is None else bias.double(), padding=2)
    xd, wd, bd = x.to(dev()), w.to(dev()), None if bias is None else bias.to(dev())
    prev = HF._CONV_MATH[0]
    try:
        HF.set_conv_math("bf16x3")
        got = HF.conv_apply(xd, wd, wd, int(dgrad), bd, B, Ci, H, W, 64, 5, False)
        HF._SCIN_MFMA[0] = False
        direct = HF.conv_apply(xd, wd, wd, int(dgrad), bd, B, Ci, H, W, 64, 5, False)
    finally:
        HF._SCIN_MFMA[0] = True
        HF.set_conv_math(prev)
    assert rel_err(direct, ref) < 1e-5
    assert rel_err(got, ref) < 5e-5
    assert not torch.equal(got, direct)      # really the other kernel


@pytest.mark.parametrize("B,S,Cs,stem", [(3, 32, 3, True), (2, 64, 3, False), (5, 32, 2, False), (2, 64, 1, True),
                                          (1, 128, 3, True), (1, 128, 3, False), (1, 256, 3, True)])
def test_wgrad5_on_planes(HF, B, S, Cs, stem):
    """5x5 weight gradient with a <= 3-channel side (stem / predict) on the bf16 matrix cores: rows (channel, dw),
    pixel reduction through the transposing LDS read -- against fp64 (bf16x3: 5e-5), accumulate form, bitwise repeatable."""
    assert HF.lib.itcv_conv2d_wgrad5_bf16p_supported(Cs, 64, S, S)
    g = torch.Generator().manual_seed(B * 10 + S + Cs)
    Ci, Co = (Cs, 64) if stem else (64, Cs)
    x = torch.randn(B, Ci, S, S, generator=g)
    dy = torch.randn(B, Co, S, S, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 5, 5), dy.double(), padding=2)
    small, big = (x, dy) if stem else (dy, x)
    bp = HF.split_planes(big.to(dev()), 2)
    dw = HF.conv_wgrad5_planes(small.to(dev()), bp, B, Cs, S, S, stem)
    assert rel_err(dw, ref) < 5e-5
    acc = HF.conv_wgrad5_planes(small.to(dev()), bp, B, Cs, S, S, stem, out=dw.clone(), accumulate=True)
    assert rel_err(acc, 2 * ref) < 5e-5
    assert torch.equal(dw, HF.conv_wgrad5_planes(small.to(dev()), bp, B, Cs, S, S, stem))


@pytest.mark.parametrize("shape", [(8, 64, 64, 64, 64), (4, 128, 32, 32, 128), (16, 64, 16, 16, 256), (64, 32, 8, 8, 64)])
@pytest.mark.parametrize("groups", [1, 2])
def test_batchnorm_statistics_from_the_conv_epilogue(HF, shape, groups):
    """conv -> BatchNorm with the statistics taken from the conv epilogue's per-tile channel sums (fp32 per 256-value
    tile, fp64 across tiles) == the statistics pass over the tensor (fp64 throughout): mean / rstd to 1e-6 of their
    scale, the normalised output to 2e-6, running buffers alike; also per BatchNorm group."""
    B, Ci, H, W, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    d = dev()
    x = torch.randn(B, Ci, H, W, generator=g).to(d)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(d)
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).to(d), torch.randn(Co, generator=g).to(d)
    HF.set_conv_math("bf16x3")
    try:
        res = {}
        for fused in (True, False):
            HF._FUSE_STATS[0] = fused
            y = HF.Conv2dFn.apply(x, w, None, False)
            assert (getattr(y, "_itcv_tile_stats", None) is not None) == (fused and bool(
                HF.lib.itcv_conv2d_fwd_bf16p_stat_tiles(B, Ci, H, W, Co, 3, 2)))
            rm, rv, nbt = torch.zeros(Co, device=d), torch.ones(Co, device=d), torch.zeros((), dtype=torch.long, device=d)
            out = HF.BnActFn.apply(y, gamma, beta, None, rm, rv, nbt, 1e-4, 0.1, 0.2, False, True, None, 2, 0, True, True, groups)
            res[fused] = (y.clone(), out.clone(), rm, rv, int(nbt), HF._tagged_planes(out, 2).clone())
    finally:
        HF._FUSE_STATS[0] = False
        HF.set_conv_math("fp32")
    a, b = res[True], res[False]
    # the conv output itself: the statistics-producing launch uses the 32x32x16 MFMA, the plain one 16x16x32 (different
    # summation order inside a 32-channel group)
    assert rel_err(a[0], b[0]) < 1e-6
    assert rel_err(a[1], b[1]) < 2e-6 and rel_err(a[2], b[2]) < 1e-6 and rel_err(a[3], b[3]) < 1e-6 and a[4] == b[4] == groups
