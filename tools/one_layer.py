#!/usr/bin/env python3
"""Runs one conv layer (forward) repeatedly in a given math mode -- a target for rocprofv3 --pmc."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
Ci, Co, S, KS, B = (int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (128, 128, 32, 3, 64)))
kind = sys.argv[7] if len(sys.argv) > 7 else "fwd"
d = torch.device("cuda:0")
x = torch.randn(B, Ci, S, S, device=d); w = torch.randn(Co, Ci, KS, KS, device=d) * 0.05; dy = torch.randn(B, Co, S, S, device=d)
HF.set_conv_math(mode)
for _ in range(20):
    if kind == "fwd":
        HF.conv_apply(x, w, w, 0, None, B, Ci, S, S, Co, KS, False)
    else:
        HF.conv_wgrad_raw(x, dy, B, Ci, S, S, Co, KS, False)
torch.cuda.synchronize()
