#!/usr/bin/env python3
"""Split-bf16 forward conv: gather kernel (fp32 input) vs LDS-DMA kernel (pre-split planes input)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF  # noqa: E402

LAYERS = [  # Ci, Co, S, KS, up2
    (64, 128, 32, 3, 0), (128, 128, 32, 3, 0), (128, 256, 16, 3, 0), (256, 256, 16, 3, 0), (256, 512, 8, 3, 0),
    (512, 512, 8, 3, 0), (512, 512, 4, 3, 0), (64, 64, 64, 3, 0), (128, 64, 32, 3, 0), (128, 64, 64, 3, 1),
    (512, 256, 16, 3, 1), (8192, 256, 1, 1, 0),
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
    B = 64
    d = torch.device("cuda:0")
    for mode, ns in (("bf16x3", 2), ("bf16x6", 3)):
        HF.set_conv_math(mode)
        tot = [0.0, 0.0, 0.0]
        for Ci, Co, S, KS, up2 in LAYERS:
            Ss = S // 2 if up2 else S
            x = torch.randn(B, Ci, Ss, Ss, device=d)
            w = torch.randn(Co, Ci, KS, KS, device=d) * 0.05
            gf = 2.0 * B * S * S * Co * Ci * KS * KS * 1e-9
            ref = HF.conv_apply(x, w, w, 0, None, B, Ci, S, S, Co, KS, bool(up2))
            xp = HF.split_planes(x, ns)
            got = HF.conv_apply_planes(xp, w, w, 0, None, B, Ci, S, S, Co, KS, bool(up2), ns)
            same = bool(torch.equal(ref, got))
            err = float((ref - got).abs().max())
            t0 = timeit(lambda: HF.conv_apply(x, w, w, 0, None, B, Ci, S, S, Co, KS, bool(up2)))
            t1 = timeit(lambda: HF.conv_apply_planes(xp, w, w, 0, None, B, Ci, S, S, Co, KS, bool(up2), ns))
            t2 = timeit(lambda: HF.split_planes(x, ns))
            print(f"{mode} {Ci:4d}->{Co:4d}@{S:3d} k{KS} up{up2} {gf:6.2f} GF | gather {t0*1e6:7.1f} us {gf/t0*1e-3:6.1f} | "
                  f"planes {t1*1e6:7.1f} us {gf/t1*1e-3:6.1f} | split {t2*1e6:6.1f} us | equal {same} maxdiff {err:.1e}", flush=True)
            tot[0] += t0; tot[1] += t1; tot[2] += t2
        print(f"{mode} sum: gather {tot[0]*1e6:.0f} us, planes {tot[1]*1e6:.0f} us, split {tot[2]*1e6:.0f} us")


if __name__ == "__main__":
    main()
