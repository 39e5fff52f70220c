import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
from hipvae import functional as HF
HF.set_conv_math("bf16x3")
d = torch.device("cuda:0")
B, Ci, Co, S, KS = 64, 128, 128, 32, 3
x = torch.randn(B, Ci, S, S, device=d); w = torch.randn(Co, Ci, KS, KS, device=d) * 0.05
xp = HF.split_planes(x, 2)
for _ in range(50):
    y = HF.conv_apply_planes(xp, w, w, 0, None, B, Ci, S, S, Co, KS, False, 2)
torch.cuda.synchronize()
v = y.flatten()[:4].tolist()
print("ablate", os.environ.get("ITCV_ABLATE"), "block0 cycles %.0f ticks %.0f -> %.2f GHz, %.1f us | block300 cycles %.0f ticks %.0f -> %.2f GHz %.1f us | ideal cycles %d" % (
    v[0], v[1], v[0] / max(v[1], 1) / 10, v[1] / 100, v[2], v[3], v[2] / max(v[3], 1) / 10, v[3] / 100, 36 * 24 * 32))
