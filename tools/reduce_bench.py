"""Micro-benchmark of the deferred weight-gradient slab fold (itcv_wgrad_reduce_many) on the c2 layer shapes: the slabs of one
backward pass are produced by the real weight-gradient kernels (random planes), the fold is timed alone with HIP events,
launched back to back.  Measured while the fold was rebuilt in round 3 (c2 shapes, 282 MB of slabs): 64 elements x 9 taps
per block with an LDS exchange 140-156 us whatever the tiling / persistence / plane padding; 4 elements x 9 taps per
thread without LDS 84 us (fine tiles for >= 64 slices; 157 us without them; 126 / 195 us with fine tiles from 16 / 8
slices).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intro-tc-vae_amd"))
from hipvae import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
HF.set_conv_math("f16x3")
# (B, Ci, H, W, Co, up2) of the encoder's 3x3 convs at c2 (batched passes: B = 128) and the decoder's
LAYERS = [(128, 64, 32, 32, 128, 0), (128, 128, 16, 16, 256, 0), (128, 256, 8, 8, 512, 0), (128, 512, 4, 4, 512, 0),
          (128, 512, 8, 8, 512, 1), (128, 512, 16, 16, 256, 1), (128, 256, 32, 32, 128, 1), (128, 128, 64, 64, 64, 1),
          (128, 64, 64, 64, 64, 0), (128, 64, 64, 64, 64, 0)]
g = torch.Generator().manual_seed(0)
ops = []
for (B, Ci, H, W, Co, up2) in LAYERS:
    Hs, Ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, Hs, Ws, generator=g).to(dev)
    dy = (torch.randn(B, Co, H, W, generator=g) * 1e-3).to(dev)
    xp, dyp = HF.split_planes(x, 4), HF.split_planes(dy, 4, gradient=True)
    dw = torch.zeros(Co, Ci, 3, 3, device=dev)
    ops.append((xp, dyp, B, Ci, H, W, Co, up2, dw))


def run():
    """Slabs of one backward, then the fold launched 20 times back to back between two events (the Python side of
    flush_wgrad_reduces takes longer than the kernel: timing one call would time the host)."""
    from hipvae.abi import call, ptr, stream
    with HF.deferred_wgrad_reduces():
        for (xp, dyp, B, Ci, H, W, Co, up2, dw) in ops:
            HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, out=dw, accumulate=True, ns=4)
        nbytes = sum(p[0].numel() for p in HF._DEFER["pending"])
        HF.flush_wgrad_reduces()
    (tab,) = HF._DEFER["tables"].values()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call("itcv_wgrad_reduce_many", ptr(tab[0]), tab[1], tab[2], stream())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20, nbytes


ts = []
for _ in range(4):
    HF._DEFER["tables"].clear()
    t, nbytes = run()
    ts.append(t)
print(f"fold {min(ts[1:]):7.1f} us (median {sorted(ts[1:])[1]:7.1f}), slabs {nbytes / 1e6:6.1f} MB -> {nbytes / 1e6 / min(ts[1:]):5.2f} TB/s of slab reads",
      flush=True)
