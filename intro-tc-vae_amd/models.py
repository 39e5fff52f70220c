"""Drop-in mirror of the reference's ``models`` module: SoftIntroVAE / Encoder / Decoder with the
same constructor signatures, attribute names, parameter initialisation order and
``state_dict`` keys as /root/reference/models.py:196-355, executing on hand-written HIP
kernels (hipvae.functional -> libitcv_hip.so).

Layer classes subclass the stock torch modules only to inherit their parameters, default
initialisation and ``state_dict`` layout; every ``forward`` is replaced.  ``Encoder`` /
``Decoder`` run a fused schedule over their children:

    conv -> [BN statistics] -> BN-apply + LeakyReLU (+ 2x2 average pool)     (one pass)
    nearest x2 upsample folded into the next convolution's im2col gather (never materialised)

while ``self.main`` keeps the reference's child names (``main.0``, ``main.res_in_16.bn1`` ...)
and still works layer by layer (``fused = False``), which is what the parity tests compare the
fused schedule against.  There is no CPU path: a CPU tensor raises in hipvae.abi.
"""
import contextlib
import math

import torch
import torch.nn as nn

from hipvae import functional as HF
from ops import reparameterize

LRELU_SLOPE = 0.2


# ------------------------------------------------------------------------------ layers
# Construction-time only: the reference sizes the encoder's fc layer by pushing a zero image through `main` on the
# CPU in train mode (models.py:229,235-238), which also warms every encoder BatchNorm's running statistics.  While
# this flag is set the layers below run their torch parent classes on the CPU so that the same pass can be made
# here (inception blocks: the biased 1x1 convs make the warm-up statistics weight dependent).  Never set in a step.
_CPU_WARMUP = [False]


# Number of independent network passes stacked in the batch of the current forward call (see bn_groups below)
_BN_GROUPS = [1]


@contextlib.contextmanager
def bn_groups(groups):
    """Inside this block a forward call treats its batch as ``groups`` independent passes of B/groups images each,
    stacked along dim 0: the convolutions / linears see one large batch (the passes share their weights), every
    BatchNorm layer normalises each pass with its OWN batch statistics and advances its running buffers once per
    pass, in order -- numerically the same as ``groups`` separate forward calls (models.py:37-38), but each conv GEMM,
    weight gradient and weight packing runs once on ``groups`` times the pixels.  The solvers use it for the passes of
    a step that the reference issues back to back on the same weights (solvers/intro.py:70-86,119-134)."""
    prev, _BN_GROUPS[0] = _BN_GROUPS[0], int(groups)
    try:
        yield
    finally:
        _BN_GROUPS[0] = prev


# Forward hooks.  The reference's anomaly detection registers an `isnan` hook on EVERY submodule (train.py:119-138).  On the
# fused schedule leaf modules are bypassed (LeakyReLU / pooling / upsampling run inside the BatchNorm and conv kernels)
# and, in the split-bf16 modes, an intermediate fp32 tensor may never be written (its consumers read planes).  So a
# network with any forward (pre-)hook registered below it runs the reference's own module-by-module sequence instead
# (`_HOOKED`): every submodule is called, in the reference's order, and every output it hands to a hook is a fully
# written fp32 tensor.  Same kernels, a few more launches.
_HOOKED = [False]


def _has_hooks(mod):
    g = nn.modules.module
    if g._global_forward_hooks or g._global_forward_pre_hooks:
        return True
    return any(m._forward_hooks or m._forward_pre_hooks for m in mod.modules())


@contextlib.contextmanager
def _hooked_scope(flag):
    prev, _HOOKED[0] = _HOOKED[0], bool(flag)
    try:
        yield
    finally:
        _HOOKED[0] = prev


def _add(a, b):
    return a + b if _CPU_WARMUP[0] else HF.AddFn.apply(a, b)


def _upsample2(x):
    return nn.functional.interpolate(x, scale_factor=2, mode="nearest") if _CPU_WARMUP[0] else HF.Upsample2Fn.apply(x)


def _avgpool2(x):
    return nn.functional.avg_pool2d(x, 2) if _CPU_WARMUP[0] else HF.AvgPool2Fn.apply(x)


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (stride 1, 'same' padding) on the implicit-GEMM MFMA kernel."""

    def forward(self, x, up2=False):
        if _CPU_WARMUP[0]:
            return nn.Conv2d.forward(self, _upsample2(x) if up2 else x)
        return HF.Conv2dFn.apply(x, self.weight, self.bias, bool(up2))


class HipLinear(nn.Linear):
    def forward(self, x):
        return HF.LinearFn.apply(x, self.weight, self.bias)


class HipBatchNorm2d(nn.BatchNorm2d):
    """BatchNorm2d with optional fused LeakyReLU / residual add / 2x2 average pool.
    ``sync_group`` (a torch.distributed group) switches the statistics to Sync-BN."""

    sync_group = None

    def forward(self, x, slope=1.0, pool=False, skip=None, out_mode=(0, True), grad_mode=(0, True)):
        """``out_mode`` / ``grad_mode`` = (planes, fp32): number of bf16 planes in which the output / the input
        gradient are additionally written for the neighbouring conv GEMMs (0 = none), and whether the fp32
        tensor itself is still needed (False: only the planes are written); see hipvae.functional."""
        if _CPU_WARMUP[0]:
            y = nn.BatchNorm2d.forward(self, x)
            y = y if skip is None else y + skip
            y = nn.functional.leaky_relu(y, slope) if slope != 1.0 else y
            return nn.functional.avg_pool2d(y, 2) if pool else y
        return HF.BnActFn.apply(x, self.weight, self.bias, skip, self.running_mean, self.running_var,
                                self.num_batches_tracked, self.eps, self.momentum, slope, bool(pool), self.training,
                                self.sync_group, int(out_mode[0]), int(grad_mode[0]), bool(out_mode[1]),
                                bool(grad_mode[1]), _BN_GROUPS[0])


class HipLeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        if _CPU_WARMUP[0]:
            return nn.functional.leaky_relu(x, self.negative_slope)
        return HF.LeakyReluFn.apply(x, self.negative_slope)


class HipAvgPool2d(nn.AvgPool2d):
    def forward(self, x):
        return _avgpool2(x)


class HipUpsample(nn.Upsample):
    def forward(self, x):
        return HF.Upsample2Fn.apply(x)


class HipSigmoid(nn.Sigmoid):
    def forward(self, x):
        return HF.SigmoidFn.apply(x)


def _conv(inc, outc, ks, bias=False):
    return HipConv2d(inc, outc, kernel_size=ks, stride=1, padding=ks // 2, groups=1, bias=bias)


def _expand(inc, outc, by_identity=False):
    """1x1 ``conv_expand`` of a block.  ``by_identity``: the reference's ConvolutionalBlock and InceptionResnetBlock
    test ``inc is not outc`` (models.py:15,149) -- object identity, so two equal widths held by DIFFERENT int objects
    (values > 256 parsed at run time, e.g. from a JSON config) still get the layer and its ``state_dict`` keys (and the
    inception block then runs it on the skip path); ResidualBlock compares values (models.py:69).  Mirrored exactly."""
    differs = (inc is not outc) if by_identity else (inc != outc)
    return _conv(inc, outc, 1) if differs else None


# ------------------------------------------------------------------------------ blocks
class ConvolutionalBlock(nn.Module):
    """models.py:8-54: (conv3x3 -> BN(eps 1e-4) -> LeakyReLU 0.2) x 2.  ``conv_expand`` exists for
    state_dict compatibility and, as in the reference, is never used by forward."""

    def __init__(self, inc=64, outc=64, groups=1, scale=1.0):
        super().__init__()
        if groups != 1:
            raise ValueError("groups != 1 is not supported by the HIP path")
        midc = int(outc * scale)
        self.eps = 1e-4
        self.conv_expand = _expand(inc, outc, by_identity=True)
        self.conv1 = _conv(inc, midc, 3)
        self.bn1 = HipBatchNorm2d(midc, eps=self.eps)
        self.relu1 = HipLeakyReLU(LRELU_SLOPE, inplace=True)
        self.conv2 = _conv(midc, outc, 3)
        self.bn2 = HipBatchNorm2d(outc, eps=self.eps)
        self.relu2 = HipLeakyReLU(LRELU_SLOPE, inplace=True)

    def forward(self, x, pool=False, up2=False, consumer=None, consumer_up2=False):
        """``consumer``: the conv module that reads this block's output (its planes are emitted by bn2)."""
        if _HOOKED[0]:      # models.py:49-54, module by module
            y = self.relu1(self.bn1(self.conv1(x, up2=up2)))
            y = self.relu2(self.bn2(self.conv2(y)))
            return _avgpool2(y) if pool else y
        B, H, W = x.size(0), x.size(2) * (2 if up2 else 1), x.size(3) * (2 if up2 else 1)
        y = self.bn1(self.conv1(x, up2=up2), slope=LRELU_SLOPE, out_mode=HF.conv_input_mode(self.conv2, B, H, W),
                     grad_mode=HF.conv_grad_mode(self.conv1, B, H, W, x.requires_grad))
        Hc, Wc = (H // 2, W // 2) if pool else (H, W)
        if consumer_up2:
            Hc, Wc = Hc * 2, Wc * 2
        return self.bn2(self.conv2(y), slope=LRELU_SLOPE, pool=pool,
                        out_mode=HF.conv_input_mode(consumer, B, Hc, Wc, consumer_up2) if consumer is not None else (0, True),
                        grad_mode=HF.conv_grad_mode(self.conv2, B, H, W))


class ResidualBlock(nn.Module):
    """models.py:57-115: conv-BN-LReLU-conv-BN, + (1x1-expanded) input, LReLU.  BN eps 1e-5."""

    def __init__(self, inc=64, outc=64, groups=1, scale=1.0):
        super().__init__()
        if groups != 1:
            raise ValueError("groups != 1 is not supported by the HIP path")
        midc = int(outc * scale)
        self.conv_expand = _expand(inc, outc)
        self.conv1 = _conv(inc, midc, 3)
        self.bn1 = HipBatchNorm2d(midc)
        self.relu1 = HipLeakyReLU(LRELU_SLOPE, inplace=True)
        self.conv2 = _conv(midc, outc, 3)
        self.bn2 = HipBatchNorm2d(outc)
        self.relu2 = HipLeakyReLU(LRELU_SLOPE, inplace=True)

    def forward(self, x, pool=False, up2=False, consumer=None, consumer_up2=False):
        if _HOOKED[0]:      # models.py:104-115, module by module
            if up2:
                x = _upsample2(x)
            identity = self.conv_expand(x) if self.conv_expand is not None else x
            y = self.relu1(self.bn1(self.conv1(x)))
            y = self.relu2(_add(self.bn2(self.conv2(y)), identity))
            return _avgpool2(y) if pool else y
        if self.conv_expand is not None:
            skip = self.conv_expand(x, up2=up2)
        else:
            skip = HF.Upsample2Fn.apply(x) if up2 else x
        # the block output also feeds the next block's skip path and x feeds this one's: fp32 stays everywhere
        y = self.bn1(self.conv1(x, up2=up2), slope=LRELU_SLOPE,
                     out_mode=(HF.conv_input_planes_ns(self.conv2), True),
                     grad_mode=(HF.conv_grad_planes_ns(self.conv1, x.requires_grad), True))
        return self.bn2(self.conv2(y), slope=LRELU_SLOPE, pool=pool, skip=skip,
                        out_mode=(HF.conv_input_planes_ns(consumer, consumer_up2), True),
                        grad_mode=(HF.conv_grad_planes_ns(self.conv2), True))


class Conv2dBatchNorm(nn.Module):
    """models.py:118-138."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding=0, groups=1):
        super().__init__()
        if stride != 1 or groups != 1 or padding != kernel_size // 2:
            raise ValueError("only stride 1 / 'same' padding / groups 1 run on the HIP path")
        self.conv = _conv(in_size, out_size, kernel_size)
        self.eps = 1e-4
        self.batch_norm = HipBatchNorm2d(out_size, eps=self.eps)
        self.relu = HipLeakyReLU(LRELU_SLOPE, inplace=True)

    def forward(self, x, up2=False):
        if _HOOKED[0]:      # models.py:134-138
            return self.relu(self.batch_norm(self.conv(x, up2=up2)))
        return self.batch_norm(self.conv(x, up2=up2), slope=LRELU_SLOPE)


class InceptionResnetBlock(nn.Module):
    """models.py:141-182: two 1x1 branches, concat, biased 1x1 conv, + (expanded) input, LReLU."""

    def __init__(self, inc=64, outc=64, groups=1, scale=1.0):
        super().__init__()
        self.eps = 1e-4
        midc = int(outc * scale)
        assert outc % 2 == 0
        self.conv_expand = _expand(inc, outc, by_identity=True)
        self.branch_0 = Conv2dBatchNorm(inc, outc // 2, kernel_size=1, stride=1, groups=groups)
        self.branch_1 = nn.Sequential(
            Conv2dBatchNorm(inc, midc, kernel_size=1, stride=1, groups=groups),
            Conv2dBatchNorm(midc, outc // 2, kernel_size=1, stride=1, groups=groups),
        )
        self.conv = _conv(outc, outc, 1, bias=True)
        self.relu = HipLeakyReLU(LRELU_SLOPE, inplace=True)

    def forward(self, x, pool=False, up2=False, consumer=None, consumer_up2=False):
        if up2:
            x = _upsample2(x)
        skip = self.conv_expand(x) if self.conv_expand is not None else x
        y = torch.cat((self.branch_0(x), self.branch_1(x)), dim=1)
        y = self.relu(_add(self.conv(y), skip))
        return _avgpool2(y) if pool else y


_BLOCKS = {"conv": ConvolutionalBlock, "res": ResidualBlock, "inception": InceptionResnetBlock}


def get_conv_class(arch):
    """models.py:185-193."""
    try:
        return _BLOCKS[arch]
    except KeyError:
        raise ValueError() from None


# ------------------------------------------------------------------------------ encoder / decoder
class Encoder(nn.Module):
    """models.py:196-244."""

    def __init__(self, arch="res", cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256):
        super().__init__()
        self.conv_block = get_conv_class(arch)
        self.zdim, self.cdim, self.image_size = zdim, cdim, image_size
        self.fused = True
        cc = channels[0]
        self.main = nn.Sequential(
            _conv(cdim, cc, 5),
            HipBatchNorm2d(cc, eps=1e-4),
            HipLeakyReLU(LRELU_SLOPE, inplace=True),
            HipAvgPool2d(2),
        )
        sz = image_size // 2
        self._stages = []  # (block name, pooled afterwards)
        for ch in channels[1:]:
            self.main.add_module(f"res_in_{sz}", self.conv_block(cc, ch, scale=1.0))
            self.main.add_module(f"down_to_{sz // 2}", HipAvgPool2d(2))
            self._stages.append((f"res_in_{sz}", True))
            cc, sz = ch, sz // 2
        self.main.add_module(f"res_in_{sz}", self.conv_block(cc, cc, scale=1.0))
        self._stages.append((f"res_in_{sz}", False))
        self.conv_output_size = torch.Size((cc, sz, sz))
        num_fc_features = cc * sz * sz
        self._warm_batchnorm(arch)
        print("conv shape: ", self.conv_output_size)
        print("num fc features: ", num_fc_features)
        self.fc = HipLinear(num_fc_features, 2 * zdim)

    def _warm_batchnorm(self, arch):
        """The reference sizes its fc layer by pushing a zero image through ``main`` in train mode
        (models.py:229,235-238), which leaves every encoder BatchNorm with running_var = 0.9 and
        num_batches_tracked = 1.  With bias-free convolutions the activations of that pass are
        identically zero, so the buffers are set directly (shape arithmetic replaces the pass)."""
        if arch == "inception":
            # biased 1x1 convs: the warm-up activations depend on the freshly initialised weights -- make the pass
            was_training = self.training
            self.train()
            _CPU_WARMUP[0] = True
            try:
                with torch.no_grad():
                    self.main(torch.zeros(1, self.cdim, self.image_size, self.image_size))
            finally:
                _CPU_WARMUP[0] = False
                self.train(was_training)
            return
        for m in self.main.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_var.fill_(0.9)
                m.num_batches_tracked.fill_(1)

    def forward(self, x):
        hooked = _has_hooks(self)
        if hooked:
            with _hooked_scope(True):
                y = self.main(x)
        elif self.fused:
            blocks = [getattr(self.main, name) for name, _ in self._stages]
            res_block = not isinstance(blocks[0], ConvolutionalBlock)    # skip paths / branches read the fp32 tensor
            first = getattr(blocks[0], "conv1", None)
            mode = HF.conv_input_mode(first, x.size(0), x.size(2) // 2, x.size(3) // 2) if first is not None else (0, True)
            y = self.main[1](self.main[0](x), slope=LRELU_SLOPE, pool=True,
                             out_mode=(mode[0], True) if res_block else mode,
                             grad_mode=(HF.conv_grad_planes_ns(self.main[0], x.requires_grad), True))
            for k, (name, pooled) in enumerate(self._stages):
                nxt = getattr(blocks[k + 1], "conv1", None) if k + 1 < len(blocks) else None
                y = blocks[k](y, pool=pooled, consumer=nxt)
        else:
            y = self.main(x)
        y = self.fc(y.reshape(x.size(0), -1))
        # models.py:243 returns the two column halves as views; every latent kernel wants dense rows, so they are made
        # contiguous ONCE here (a view would be copied again by each of the 3-6 ops that consume it)
        mu, logvar = y.chunk(2, dim=1)
        return mu.contiguous(), logvar.contiguous()


class Decoder(nn.Module):
    """models.py:247-298."""

    def __init__(self, arch="res", cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256,
                 conv_input_size=None):
        super().__init__()
        self.conv_block = get_conv_class(arch)
        self.cdim, self.image_size = cdim, image_size
        self.fused = True
        cc = channels[-1]
        self.conv_input_size = conv_input_size
        num_fc_features = cc * 4 * 4 if conv_input_size is None else int(math.prod(conv_input_size))
        if conv_input_size is None:
            self.conv_input_size = torch.Size((cc, 4, 4))
        self.fc = nn.Sequential(HipLinear(zdim, num_fc_features), HipLeakyReLU(LRELU_SLOPE, inplace=True))
        sz = int(math.sqrt(num_fc_features // cc))
        self.main = nn.Sequential()
        self._stages = []
        for ch in channels[::-1]:
            self.main.add_module(f"res_in_{sz}", self.conv_block(cc, ch, scale=1.0))
            self.main.add_module(f"up_to_{sz * 2}", HipUpsample(scale_factor=2, mode="nearest"))
            self._stages.append(f"res_in_{sz}")
            cc, sz = ch, sz * 2
        self.main.add_module(f"res_in_{sz}", self.conv_block(cc, cc, scale=1.0))
        self._stages.append(f"res_in_{sz}")
        self.main.add_module("predict", _conv(cc, cdim, 5, bias=True))
        self.main.add_module("sigmoid", HipSigmoid())

    def forward(self, z):
        z = z.reshape(z.size(0), -1)
        y = self.fc(z).view(z.size(0), *self.conv_input_size)
        if _has_hooks(self):
            with _hooked_scope(True):
                y = self.main(y)
        elif self.fused:
            blocks = [getattr(self.main, name) for name in self._stages]
            for k, blk in enumerate(blocks):   # the upsample before block k folds into its conv
                nxt = getattr(blocks[k + 1], "conv1", None) if k + 1 < len(blocks) else self.main.predict
                y = blk(y, up2=k > 0, consumer=nxt, consumer_up2=k + 1 < len(blocks))
            y = self.main.sigmoid(self.main.predict(y))
        else:
            y = self.main(y)
        return y


class SoftIntroVAE(nn.Module):
    """models.py:301-355."""

    def __init__(self, arch="res", cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256):
        super().__init__()
        self.zdim: int = zdim
        self.cdim: int = cdim
        self.encoder = Encoder(arch, cdim, zdim, channels, image_size)
        self.decoder = Decoder(arch, cdim, zdim, channels, image_size,
                               conv_input_size=self.encoder.conv_output_size)

    def forward(self, x, deterministic=False):
        mu, logvar = self.encode(x)
        z = mu if deterministic else reparameterize(mu, logvar)
        return mu, logvar, z, self.decode(z)

    def sample(self, z):
        return self.decode(z)

    def sample_with_noise(self, num_samples=1, device=torch.device("cpu")):
        return self.decode(torch.randn(num_samples, self.zdim).to(device))

    def encode(self, x):
        return self.encoder(x)

    def decode(self, z):
        return self.decoder(z)

    def set_fused(self, flag):
        self.encoder.fused = self.decoder.fused = bool(flag)
        return self
