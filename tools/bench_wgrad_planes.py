#!/usr/bin/env python3
"""Weight gradient: split-in-kernel (fp32 inputs) vs planes + transposing LDS reads."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF  # noqa: E402

LAYERS = [  # Ci, Co, S, up2
    (64, 128, 32, 0), (128, 128, 32, 0), (128, 256, 16, 0), (256, 256, 16, 0), (256, 512, 8, 0),
    (512, 512, 8, 0), (512, 512, 4, 0), (64, 64, 64, 0), (128, 64, 32, 0), (128, 64, 64, 1), (512, 256, 16, 1),
    (512, 512, 8, 1),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    B, KS = int(os.environ.get("B", 64)), 3
    d = torch.device("cuda:0")
    tot = [0.0, 0.0]
    for Ci, Co, S, up2 in LAYERS:
        Ss = S // 2 if up2 else S
        x = torch.randn(B, Ci, Ss, Ss, device=d)
        dy = torch.randn(B, Co, S, S, device=d)
        gf = 2.0 * B * S * S * Co * Ci * KS * KS * 1e-9
        HF.set_conv_math("fp32")
        ref = HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, bool(up2))
        HF.set_conv_math("bf16x3")
        old = HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, bool(up2))
        t0 = timeit(lambda: HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, bool(up2)))
        line = f"{Ci:4d}->{Co:4d}@{S:3d} up{up2} {gf:6.2f} GF | old {t0*1e6:7.1f} us {gf/t0*1e-3:6.1f} err {float((old-ref).abs().max()/ref.abs().max()):.1e}"
        if HF.lib.itcv_conv2d_wgrad_bf16p_supported(B, Ci, S, S, Co, KS):
            xp, dyp = HF.split_planes(x, 2), HF.split_planes(dy, 2)
            got = HF.conv_wgrad_planes(xp, dyp, B, Ci, S, S, Co, KS, bool(up2))
            err = float((got - ref).abs().max() / ref.abs().max())
            acc = HF.conv_wgrad_planes(xp, dyp, B, Ci, S, S, Co, KS, bool(up2), out=got.clone(), accumulate=True)
            err2 = float((acc - 2 * got).abs().max() / ref.abs().max())
            t1 = timeit(lambda: HF.conv_wgrad_planes(xp, dyp, B, Ci, S, S, Co, KS, bool(up2)))
            line += f" | planes {t1*1e6:7.1f} us {gf/t1*1e-3:6.1f} err {err:.1e} acc {err2:.1e}"
            tot[1] += t1
        else:
            tot[1] += t0
        tot[0] += t0
        print(line, flush=True)
    print(f"sum: old {tot[0]*1e6:.0f} us, planes {tot[1]*1e6:.0f} us")


if __name__ == "__main__":
    main()
