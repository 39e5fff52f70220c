"""Diagnostic: in-kernel main-loop cycles of the band conv kernel (ITCV_ABLATE=64) vs the MFMA-issue ideal."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF
HF.set_conv_math("bf16x3")
d = torch.device("cuda:0")
for Ci, Co, S in ((64, 64, 64), (128, 128, 32), (64, 128, 32), (512, 512, 8)):
    B, KS = 64, 3
    x = torch.randn(B, Ci, S, S, device=d); w = torch.randn(Co, Ci, KS, KS, device=d) * 0.05
    xp = HF.split_planes(x, 2)
    for _ in range(30):
        y = HF.conv_apply_planes(xp, w, w, 0, None, B, Ci, S, S, Co, KS, False, 2)
    torch.cuda.synchronize()
    v = y.flatten()[:8].tolist()
    per = 24 * 32 * (1 if Co <= 64 else 2)   # MFMA cycles per K-tile per SIMD (2 waves)
    for o in (0, 4):
        nk = v[o + 3]
        print(f"{Ci}->{Co}@{S}: loop {v[o]:.0f} cyc = {v[o]/max(nk,1):.0f}/K-tile (ideal {per}), {v[o]/max(v[o+1],1)/10:.2f} GHz, {v[o+1]/100:.1f} us; prologue {v[o+2]:.0f} cyc; K-tiles {nk:.0f}")
