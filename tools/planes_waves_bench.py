"""A/B of 4 vs 8 MFMA waves in the 128-pixel planes kernel's 128 x 128 tile (option planes_mfma_waves) on the 4x4 layers of
c2: forward and data-gradient, results against each other, kernel + split-K reduce time (20 launches).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intro-tc-vae_amd"))
from hipvae import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
HF.set_conv_math("f16x3")
g = torch.Generator().manual_seed(0)
for (B, Ci, H, W, Co, up2) in [(128, 512, 4, 4, 512, 0), (64, 512, 4, 4, 512, 0), (128, 512, 4, 4, 256, 0), (128, 256, 4, 4, 512, 1)]:
    Hs, Ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Ci, Hs, Ws, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(dev)
    xp = HF.split_planes(x, 4)
    res = {}
    for nw in (4, 8):
        HF.set_option("planes_mfma_waves", nw)
        y = HF.conv_apply_planes(xp, w, w, 0, None, B, Ci, H, W, Co, 3, up2, 4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            HF.conv_apply_planes(xp, w, w, 0, None, B, Ci, H, W, Co, 3, up2, 4)
        e1.record()
        torch.cuda.synchronize()
        res[nw] = (y, e0.elapsed_time(e1) * 1e3 / 20)
    HF.set_option("planes_mfma_waves", 8)
    flop = 2.0 * B * H * W * Co * Ci * 9
    print(f"B={B} {Ci}->{Co} @{H}x{W} up2={up2}: 4 waves {res[4][1]:6.1f} us ({flop / res[4][1] * 1e-6:4.0f} TF)  8 waves {res[8][1]:6.1f} us "
          f"({flop / res[8][1] * 1e-6:4.0f} TF)  equal {torch.equal(res[4][0], res[8][0])}", flush=True)
