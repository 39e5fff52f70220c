"""Diagnostic: how many conv operands per step come from producer-attached planes vs an on-demand split."""
import contextlib, io, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
import models
from hipvae import functional as HF
from solvers.intro_tc import IntroTCSovler


class _DS:
    def __len__(self):
        return 10000


dev = torch.device("cuda:0")
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = models.SoftIntroVAE(arch="conv", cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)
model = model.to(dev).train()
oe = torch.optim.Adam(model.encoder.parameters(), lr=2e-4)
od = torch.optim.Adam(model.decoder.parameters(), lr=2e-4)
s = IntroTCSovler(_DS(), model, 16, oe, od, "mse", 0.5, 0.75, 512, 1e-8, dev, True, None, clip=100.0)
x = torch.rand(16, 3, 64, 64, device=dev)
s.train_step(x, 0)
HF.PLANES_STATS[:] = [0, 0]
print(s.train_step(x, 1))
print("planes: attached %d, split on demand %d" % tuple(HF.PLANES_STATS))
