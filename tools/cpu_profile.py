"""Diagnostic: where the host time of an eager step goes (cProfile over a few steps)."""
import contextlib, cProfile, io, os, pstats, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
import models
from solvers.intro_tc import IntroTCSovler
class _DS:
    def __len__(self): return 10000
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    m = models.SoftIntroVAE(arch="conv", cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)
m = m.to(dev).train()
oe = torch.optim.Adam(m.encoder.parameters(), lr=2e-4); od = torch.optim.Adam(m.decoder.parameters(), lr=2e-4)
s = IntroTCSovler(_DS(), m, 64, oe, od, "mse", 0.5, 0.75, 512, 1e-8, dev, True, None, clip=100.0)
x = torch.rand(64, 3, 64, 64, device=dev)
for i in range(3): s.train_step(x, i)
pr = cProfile.Profile(); pr.enable()
for i in range(5): s.train_step(x, i)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
