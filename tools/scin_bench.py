"""Time the <= 3-channel 5x5 conv (stem forward / predict data-gradient): direct fp32 kernel vs the bf16x3 MFMA form.

    python tools/scin_bench.py [B] [S]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intro-tc-vae_amd"))
import torch

from hipvae import functional as HF


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    HF.set_conv_math("bf16x3")
    for dgrad in (0, 1):
        x = torch.randn(B, 3, S, S, device=dev)
        w = torch.randn((3, 64, 5, 5) if dgrad else (64, 3, 5, 5), device=dev) / 6
        for mfma in (False, True):
            HF._SCIN_MFMA[0] = mfma
            for _ in range(3):
                HF.conv_apply(x, w, w, dgrad, None, B, 3, S, S, 64, 5, False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                HF.conv_apply(x, w, w, dgrad, None, B, 3, S, S, 64, 5, False)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 20
            out_gb = B * 64 * S * S * 4 / 1e9
            print(f"B={B} S={S} dgrad={dgrad} mfma={mfma}: {us:7.1f} us  ({out_gb / (us * 1e-6) / 1e3:.2f} TB/s of output)")


if __name__ == "__main__":
    main()
