#!/usr/bin/env python3
"""Time of the TC estimator (forward + backward of the live path) at the single-GPU (64 x 64 x 128) and the 8-GPU
(64 local rows x 512 global columns x 128) sizes, and the c5 size (32 x 256 x 512)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
import torch  # noqa: E402

import ops  # noqa: E402

dev = torch.device("cuda:0")
for Bl, Bt, D in ((64, 64, 128), (64, 512, 128), (128, 128, 256), (32, 256, 512)):
    g = torch.Generator().manual_seed(1)
    mu = torch.randn(Bt, D, generator=g).to(dev)
    lv = (-3 + 2 * torch.randn(Bl, D, generator=g)).to(dev).requires_grad_(True)
    z = (mu[:Bl] + torch.randn(Bl, D, generator=g).to(dev) * (0.5 * lv.detach()).exp()).requires_grad_(True)
    mu_all = mu.clone().requires_grad_(True)
    w = torch.randn(Bl, generator=g).to(dev)

    def once():
        tc = ops.total_correlation(z, mu_all[:Bl], lv, 10000, "none", mu_all=mu_all, row_offset=0)
        (w * tc).sum().backward()

    for _ in range(5):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        once()
    for _ in range(3):
        graph.replay()
    e0.record()
    for _ in range(50):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"TC fwd+bwd  rows {Bl:4d} x cols {Bt:4d} x D {D:4d}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per call (graph replay, incl. torch glue)")
