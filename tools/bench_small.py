#!/usr/bin/env python3
"""5x5 small-channel convs (stem 3->64, predict 64->3): forward / data-gradient; VALU kernels vs the planes MFMA form."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF
import torch.nn.functional as F

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

d = torch.device("cuda:0"); B = int(os.environ.get("B", 64))
for S in (64, 16):
  for name, Ci, Co, dgrad in (("stem fwd 3->64", 3, 64, 0), ("predict dgrad 3<-64 (Cin=3)", 3, 64, 1),
                            ("predict fwd 64->3", 64, 3, 0), ("stem dgrad 64->3 (Cout=3)", 64, 3, 1)):
    x = torch.randn(B, Ci, S, S, device=d)
    w = torch.randn((Ci, Co, 5, 5) if dgrad else (Co, Ci, 5, 5), device=d) * 0.05
    bias = torch.randn(Co, device=d) if not dgrad else None
    ref = F.conv2d(x.double(), (w.flip(2, 3).transpose(0, 1) if dgrad else w).double(), None if bias is None else bias.double(), padding=2)
    got = HF.conv_apply(x, w, w, dgrad, bias, B, Ci, S, S, Co, 5, False)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: HF.conv_apply(x, w, w, dgrad, bias, B, Ci, S, S, Co, 5, False))
    line = f"S={S} {name:32s} {t:7.1f} us  {2.0*B*S*S*Ci*Co*25/t*1e-6:6.1f} TF  err {err:.1e}"
    if HF.lib.itcv_conv2d_small_cout_bf16p_supported(Ci, Co, 5):
        xp = HF.split_planes(x, 2)
        g2 = HF.conv_apply_planes(xp, w, w, dgrad, bias, B, Ci, S, S, Co, 5, False, 2)
        e2 = float((g2.double() - ref).abs().max() / ref.abs().max())
        t2 = timeit(lambda: HF.conv_apply_planes(xp, w, w, dgrad, bias, B, Ci, S, S, Co, 5, False, 2))
        line += f" | planes MFMA {t2:7.1f} us err {e2:.1e}"
    print(line, flush=True)
