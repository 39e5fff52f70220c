#!/usr/bin/env python3
"""Per-layer timing of the implicit-GEMM conv kernels at the 64x64 (c2) shapes: forward,
data-gradient, weight-gradient, with achieved TFLOP/s against the 157.3 TF fp32-MFMA peak."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF  # noqa: E402

LAYERS = [  # Ci, Co, S, KS
    (3, 64, 64, 5), (64, 128, 32, 3), (128, 128, 32, 3), (128, 256, 16, 3), (256, 256, 16, 3),
    (256, 512, 8, 3), (512, 512, 8, 3), (512, 512, 4, 3), (64, 64, 64, 3), (128, 64, 32, 3), (64, 3, 64, 5),
]


def timeit(fn, iters=10):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    d = torch.device("cuda:0")
    print(f"{'layer':>22} {'GF':>7} | {'fwd ms':>8} {'TF':>6} | {'dgrad ms':>8} {'TF':>6} | {'wgrad ms':>8} {'TF':>6}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for Ci, Co, S, KS in LAYERS:
        x = torch.randn(B, Ci, S, S, device=d)
        w = torch.randn(Co, Ci, KS, KS, device=d) * 0.05
        dy = torch.randn(B, Co, S, S, device=d)
        wp, wpt = HF.pack_weight(w, 0), HF.pack_weight(w, 1)
        gf = 2.0 * B * S * S * Co * Ci * KS * KS * 1e-9
        tf = timeit(lambda: HF.conv_fwd_raw(x, wp, None, B, Ci, S, S, Co, KS, False))
        td = timeit(lambda: HF.conv_fwd_raw(dy, wpt, None, B, Co, S, S, Ci, KS, False))
        tw = timeit(lambda: HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, False))
        extra = ""
        if HF.lib.itcv_conv2d_bf16s_supported(Ci, Co, KS) and HF.lib.itcv_conv2d_bf16s_supported(Co, Ci, KS):
            ref = HF.conv_fwd_raw(x, wp, None, B, Ci, S, S, Co, KS, False)
            for mode in ("bf16x3", "bf16x6"):
                HF.set_conv_math(mode)
                t3 = timeit(lambda: HF.conv_apply(x, w, w, 0, None, B, Ci, S, S, Co, KS, False))
                err = float((HF.conv_apply(x, w, w, 0, None, B, Ci, S, S, Co, KS, False) - ref).abs().max() / ref.abs().max())
                extra += f" | {mode} {t3*1e3:6.3f} ms {gf/t3*1e-3:6.1f} TFeq err {err:.1e}"
                if HF.lib.itcv_conv2d_wgrad_bf16s_supported(Ci, S, S, Co, KS):
                    refw = None
                    HF.set_conv_math("fp32"); refw = HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, False); HF.set_conv_math(mode)
                    tw3 = timeit(lambda: HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, False))
                    errw = float((HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, False) - refw).abs().max() / refw.abs().max())
                    extra += f" wg {tw3*1e3:6.3f} ms {gf/tw3*1e-3:6.1f} err {errw:.1e}"
            HF.set_conv_math("fp32")
        print(f"{Ci:4d}->{Co:4d}@{S:3d} k{KS} B{B:3d} {gf:7.2f} | {tf*1e3:8.3f} {gf/tf*1e-3:6.1f} | "
              f"{td*1e3:8.3f} {gf/td*1e-3:6.1f} | {tw*1e3:8.3f} {gf/tw*1e-3:6.1f}" + extra)
        tot[0] += gf; tot[1] += tf; tot[2] += td; tot[3] += tw
    print(f"{'sum':>22} {tot[0]:7.2f} | {tot[1]*1e3:8.3f} {tot[0]/tot[1]*1e-3:6.1f} | {tot[2]*1e3:8.3f} "
          f"{tot[0]/tot[2]*1e-3:6.1f} | {tot[3]*1e3:8.3f} {tot[0]/tot[3]*1e-3:6.1f}")


if __name__ == "__main__":
    main()
