"""A/B of the two MFMA shapes in the planes weight-gradient kernel (option wgrad_m16) on the c2 layer shapes: results
against each other and against fp64 on a small sample, and the kernel time of each (20 launches, immediate reduce
included).  GPU box only."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intro-tc-vae_amd"))
from hipvae import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
LAYERS = [(128, 64, 64, 64, 64, 0), (128, 128, 64, 64, 64, 1), (128, 64, 32, 32, 128, 0), (128, 128, 16, 16, 256, 0),
          (128, 256, 8, 8, 512, 0), (128, 512, 4, 4, 512, 0), (128, 512, 8, 8, 512, 1), (128, 512, 16, 16, 256, 1),
          (128, 256, 32, 32, 128, 1), (64, 256, 8, 8, 512, 0), (64, 512, 4, 4, 512, 0)]
g = torch.Generator().manual_seed(0)
tot = {0: 0.0, 1: 0.0}
for ns, name in ((4, "f16x3"), (2, "bf16x3")):
    HF.set_conv_math(name)
    for (B, Ci, H, W, Co, up2) in LAYERS:
        Hs, Ws = (H // 2, W // 2) if up2 else (H, W)
        x = torch.randn(B, Ci, Hs, Ws, generator=g).to(dev)
        dy = (torch.randn(B, Co, H, W, generator=g) * 1e-3).to(dev)
        xp, dyp = HF.split_planes(x, ns), HF.split_planes(dy, ns, gradient=True)
        res = {}
        for m16 in (0, 1):
            HF.set_option("wgrad_m16", m16)
            dw = HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, ns=ns)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                HF.conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, 3, up2, ns=ns)
            e1.record()
            torch.cuda.synchronize()
            res[m16] = (dw, e0.elapsed_time(e1) * 1e3 / 20)
            if ns == 4:
                tot[m16] += res[m16][1]
        HF.set_option("wgrad_m16", 0)
        a, b = res[0][0].double(), res[1][0].double()
        err = float((a - b).abs().max() / a.abs().max())
        # fp64 on 4 output channels
        xs = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
        ref = torch.nn.grad.conv2d_weight(xs.cpu(), (4, Ci, 3, 3), dy[:, :4].double().cpu(), padding=1)
        e64 = float((b[:4].cpu() - ref).abs().max() / ref.abs().max())
        flop = 2.0 * B * H * W * Co * Ci * 9
        print(f"{name} B={B} {Ci}->{Co} @{H}x{W} up2={up2}: 32x32x16 {res[0][1]:7.1f} us ({flop / res[0][1] * 1e-6:5.0f} TF)  16x16x32 "
              f"{res[1][1]:7.1f} us ({flop / res[1][1] * 1e-6:5.0f} TF)  diff {err:.1e}  m16 vs fp64 {e64:.1e}", flush=True)
print("sum f16x3:", tot)
