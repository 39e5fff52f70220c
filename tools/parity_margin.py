#!/usr/bin/env python3
"""Prints the relative deviation of the returned losses of one intro-TC step (64x64x3, z=128, B=8) from
the CPU oracle for each conv math mode, and of the golden tiny-model steps from the reference."""
import os, sys, io, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "intro-tc-vae_amd"), ROOT]
import models, ops
from hipvae import functional as HF
from solvers.intro_tc import IntroTCSovler
from oracle.network import Net
from oracle.steps import Trainer

class DS:
    def __len__(self): return 10000

cfg = dict(cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)
B = 8
x = torch.rand(B, 3, 64, 64, generator=torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(1234)
draws = [[torch.randn(B, 128, generator=g) for _ in range(6)] for _ in range(2)]
def fresh():
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        return models.SoftIntroVAE(arch="conv", **cfg)
m = fresh(); sd = {k: v.clone() for k, v in m.state_dict().items()}
tr = Trainer("intro_tc", Net("conv", state=sd, **cfg), dataset_size=10000, beta_kl=0.5, beta_rec=0.75, beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4)
ref = [tr.step(x, d) for d in draws]
for mode in ("fp32", "bf16x6", "bf16x3"):
    HF.set_conv_math(mode)
    model = fresh().cuda().train()
    s = IntroTCSovler(DS(), model, B, torch.optim.Adam(model.encoder.parameters(), lr=2e-4), torch.optim.Adam(model.decoder.parameters(), lr=2e-4), "mse", 0.5, 0.75, 512.0, 1e-8, torch.device("cuda:0"), False, None, clip=100.0)
    s.conv_math = mode
    for i, d in enumerate(draws):
        with ops.noise_queue(d):
            got = s.train_step(x, i)
        print(mode, "step", i, {k: f"{abs(got[k]-ref[i][k])/abs(ref[i][k]):.1e}" for k in got})
