"""Oracle restatement of the reference's models.py as a functional layer program over a
flat state dict (test infrastructure, see oracle/__init__).

The network is described once as a list of primitive steps (``plan``) and interpreted by
``run``; parameters and BatchNorm buffers live in a ``dict[str, Tensor]`` keyed exactly
like the reference's ``state_dict()`` (e.g. ``encoder.main.res_in_16.bn1.running_var``).

Reference: /root/reference/models.py:8-54 (ConvolutionalBlock), :57-115 (ResidualBlock),
:118-182 (Conv2dBatchNorm / InceptionResnetBlock), :196-244 (Encoder), :247-298 (Decoder),
:301-355 (SoftIntroVAE).
"""
import math

import torch
import torch.nn.functional as F

SLOPE = 0.2
BN_MOMENTUM = 0.1


def _block_steps(kind, prefix, inc, outc):
    """One encoder/decoder block as primitive steps (models.py:8-182)."""
    s = []
    expand = inc != outc
    if kind == "conv":
        # conv_expand (models.py:15-26) is created but never used by forward (:51-54)
        s += [("conv", prefix + "conv1", 3, 1, False), ("bn", prefix + "bn1", 1e-4), ("lrelu",),
              ("conv", prefix + "conv2", 3, 1, False), ("bn", prefix + "bn2", 1e-4), ("lrelu",)]
    elif kind == "res":
        s += [("push",)]
        s += [("conv", prefix + "conv1", 3, 1, False), ("bn", prefix + "bn1", 1e-5), ("lrelu",),
              ("conv", prefix + "conv2", 3, 1, False), ("bn", prefix + "bn2", 1e-5)]
        s += [("add_skip", prefix + "conv_expand" if expand else None), ("lrelu",)]
    elif kind == "inception":
        s += [("push",),
              ("inception", prefix, expand),
              ("lrelu",)]
    else:
        raise ValueError(kind)
    return s


def plan(arch, cdim, zdim, channels, image_size):
    """Encoder / decoder step lists + the parameter/buffer shapes (models.py:196-298)."""
    channels = list(channels)
    enc = [("conv", "encoder.main.0", 5, 2, False), ("bn", "encoder.main.1", 1e-4), ("lrelu",), ("avgpool",)]
    cc, sz = channels[0], image_size // 2
    for ch in channels[1:]:
        enc += _block_steps(arch, f"encoder.main.res_in_{sz}.", cc, ch) + [("avgpool",)]
        cc, sz = ch, sz // 2
    enc += _block_steps(arch, f"encoder.main.res_in_{sz}.", cc, cc)
    conv_shape = (cc, sz, sz)
    nfeat = cc * sz * sz
    enc += [("flatten",), ("linear", "encoder.fc")]

    dec = [("linear", "decoder.fc.0"), ("lrelu",), ("view", conv_shape)]
    cc = channels[-1]
    sz = int(math.sqrt(nfeat // cc))
    for ch in channels[::-1]:
        dec += _block_steps(arch, f"decoder.main.res_in_{sz}.", cc, ch) + [("upsample",)]
        cc, sz = ch, sz * 2
    dec += _block_steps(arch, f"decoder.main.res_in_{sz}.", cc, cc)
    dec += [("conv", "decoder.main.predict", 5, 2, True), ("sigmoid",)]
    return {"encoder": enc, "decoder": dec, "conv_shape": conv_shape, "nfeat": nfeat}


def _bn(sd, key, x, eps, train):
    if train:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], training=train,
                        momentum=BN_MOMENTUM, eps=eps)


def _conv_bn_act(sd, prefix, x, train):
    """models.py:118-138 Conv2dBatchNorm (1x1 conv, BN eps 1e-4, LeakyReLU 0.2)."""
    y = F.conv2d(x, sd[prefix + "conv.weight"])
    y = _bn(sd, prefix + "batch_norm", y, 1e-4, train)
    return F.leaky_relu(y, SLOPE)


def run(steps, sd, x, train=True):
    """Interpret a step list.  BatchNorm running buffers in ``sd`` are updated in place when
    ``train`` (momentum 0.1, unbiased running variance), exactly like nn.BatchNorm2d."""
    stack = []
    for st in steps:
        op = st[0]
        if op == "conv":
            _, key, ks, pad, bias = st
            x = F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"] if bias else None, padding=pad)
        elif op == "bn":
            x = _bn(sd, st[1], x, st[2], train)
        elif op == "lrelu":
            x = F.leaky_relu(x, SLOPE)
        elif op == "avgpool":
            x = F.avg_pool2d(x, 2)
        elif op == "upsample":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif op == "sigmoid":
            x = torch.sigmoid(x)
        elif op == "flatten":
            x = x.reshape(x.shape[0], -1)
        elif op == "view":
            x = x.reshape(x.shape[0], *st[1])
        elif op == "linear":
            x = F.linear(x, sd[st[1] + ".weight"], sd[st[1] + ".bias"])
        elif op == "push":
            stack.append(x)
        elif op == "add_skip":
            skip = stack.pop()
            if st[1] is not None:
                skip = F.conv2d(skip, sd[st[1] + ".weight"])
            x = x + skip
        elif op == "inception":
            _, prefix, expand = st
            inp = stack.pop()
            # models.py:149 tests ``inc is not outc`` (identity): the layer may exist, and then runs, for equal widths
            expand = (prefix + "conv_expand.weight") in sd
            skip = F.conv2d(inp, sd[prefix + "conv_expand.weight"]) if expand else inp
            b0 = _conv_bn_act(sd, prefix + "branch_0.", inp, train)
            b1 = _conv_bn_act(sd, prefix + "branch_1.0.", inp, train)
            b1 = _conv_bn_act(sd, prefix + "branch_1.1.", b1, train)
            y = torch.cat((b0, b1), dim=1)
            x = F.conv2d(y, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"]) + skip
        else:
            raise ValueError(op)
    return x


class Net:
    """Functional SoftIntroVAE (models.py:301-355) over a state dict."""

    def __init__(self, arch, cdim, zdim, channels, image_size, state):
        self.arch, self.cdim, self.zdim, self.image_size = arch, cdim, zdim, image_size
        self.plan = plan(arch, cdim, zdim, channels, image_size)
        self.sd = state
        self.train = True

    @staticmethod
    def is_param(key):
        return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))

    def param_keys(self, part):
        return [k for k in self.sd if k.startswith(part + ".") and self.is_param(k)]

    def encode(self, x):
        y = run(self.plan["encoder"], self.sd, x, self.train)
        mu, logvar = y.chunk(2, dim=1)
        return mu, logvar

    def decode(self, z):
        return run(self.plan["decoder"], self.sd, z.reshape(z.shape[0], -1), self.train)
