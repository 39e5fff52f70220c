import contextlib, io, os, sys, time, torch
sys.path.insert(0, "intro-tc-vae_amd")
import models
from solvers.intro_tc import IntroTCSovler
class _DS:
    def __len__(self): return 10000
dev = torch.device("cuda:0")
for size, zdim, ch, B in ((256, 512, (64,128,256,512,512,512), 16), (128, 256, (64,128,256,512,512), 64)):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.SoftIntroVAE(arch="conv", cdim=3, zdim=zdim, channels=ch, image_size=size)
    m = m.to(dev).train()
    oe = torch.optim.Adam(m.encoder.parameters(), lr=2e-4); od = torch.optim.Adam(m.decoder.parameters(), lr=2e-4)
    s = IntroTCSovler(_DS(), m, B, oe, od, "mse", 0.5, 0.75, 512, 1e-8, dev, True, None, clip=100.0)
    s.enable_graph()
    x = torch.rand(B, 3, size, size, device=dev)
    for i in range(6): d = s.train_step(x, i)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(5): d = s.train_step(x, i)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(size, zdim, B, d, f"{dt*1e3:.1f} ms/step {B/dt:.0f} img/s", flush=True)
    del s, m, oe, od
    torch.cuda.empty_cache()
