"""ms/step of the intro-tc step for the three reference architectures at the c2 shape (64x64x3, z=128, B=64),
hipGraph, bf16x3.   python3 tools/arch_bench.py"""
import contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intro-tc-vae_amd"))
import models
from solvers.intro_tc import IntroTCSovler


class _DS:
    def __len__(self):
        return 10000


dev = torch.device("cuda:0")
for arch in ("conv", "res", "inception"):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.SoftIntroVAE(arch=arch, cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64).to(dev).train()
    s = IntroTCSovler(_DS(), m, 64, torch.optim.Adam(m.encoder.parameters(), lr=2e-4),
                      torch.optim.Adam(m.decoder.parameters(), lr=2e-4), "mse", 0.5, 0.75, 512.0, 1e-8, dev, True, None,
                      clip=100.0)
    s.enable_graph()
    x = torch.rand(64, 3, 64, 64, device=dev)
    for i in range(6):
        d = s.train_step(x, i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(10):
        d = s.train_step(x, i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    n = sum(p.numel() for p in m.parameters())
    print(f"{arch}: {n/1e6:.1f} M params, {dt*1e3:.1f} ms/step, {64/dt:.0f} img/s, graph={s._graph is not None}, {d}", flush=True)
    del s, m
    torch.cuda.empty_cache()
