#!/usr/bin/env python3
"""Time of one conv-BN unit's BatchNorm launches (forward statistics + apply, backward sums + apply, planes in and out)
at the shapes of the 64x64 configuration, per plane-store form (ITCV_BN_STRIP mask; run once per value)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
import torch  # noqa: E402

from hipvae import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
print("ITCV_BN_STRIP =", os.environ.get("ITCV_BN_STRIP", "(default)"))
for B, C, H, W, pool in ((64, 64, 64, 64, False), (64, 128, 32, 32, False), (64, 256, 16, 16, False), (64, 64, 64, 64, True),
                         (64, 512, 8, 8, False)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g).to(dev).requires_grad_(True)
    gamma, beta = torch.ones(C, device=dev).requires_grad_(True), torch.zeros(C, device=dev).requires_grad_(True)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    oshape = (B, C, H // 2, W // 2) if pool else (B, C, H, W)
    dy = torch.randn(*oshape, generator=g).to(dev)

    def fwd():
        return HF.BnActFn.apply(x, gamma, beta, None, rm, rv, nbt, 1e-4, 0.1, 0.2, pool, True, None, 2, 2, False, False, 1)

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        # flush caches between repetitions so the numbers are HBM numbers: a 512 MiB fill
        junk = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
        tot = 0.0
        for _ in range(n):
            junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / n * 1e3

    tf = timeit(lambda: fwd())
    y = fwd()
    tb = timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
    mb = B * C * H * W * 4 / 1e6
    print(f"B{B} C{C} {H}x{W} pool={int(pool)}: fwd {tf:7.1f} us ({12 * mb / 4 / tf:5.2f} TB/s-eq of 12 B/elem)   "
          f"bwd {tb:7.1f} us ({20 * mb / 4 / tb:5.2f} TB/s-eq of 20 B/elem)")
