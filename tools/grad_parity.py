#!/usr/bin/env python3
"""Who is right when the HIP step and the fp32 oracle disagree on a gradient tensor?  Runs one intro-TC step at the
c2 shape (B=8) three ways -- HIP (chosen conv arithmetic), CPU oracle fp32, CPU oracle fp64 -- and lists, per phase,
the gradient tensors with the largest error against fp64, relative to the phase's largest gradient tensor.

    python tools/grad_parity.py [fp32|bf16x6|bf16x3] [B]
"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "intro-tc-vae_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import models  # noqa: E402
from oracle.network import Net  # noqa: E402
from oracle.steps import Trainer  # noqa: E402
from solvers.intro_tc import IntroTCSovler  # noqa: E402
from step_trace import traced_hip_step, traced_oracle_step  # noqa: E402

math = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C2 = dict(cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)
dev = torch.device("cuda:0")
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = models.SoftIntroVAE(arch="conv", **C2)
sd = {k: v.clone() for k, v in model.state_dict().items()}
model = model.to(dev).train()


class DS:
    def __len__(self):
        return 10000


solver = IntroTCSovler(DS(), model, B, torch.optim.Adam(model.encoder.parameters(), lr=2e-4),
                       torch.optim.Adam(model.decoder.parameters(), lr=2e-4), "mse", 0.5, 0.75, 512.0, 1e-8, dev, False,
                       None, clip=100.0)
solver.conv_math = math
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, 3, 64, 64, generator=torch.Generator().manual_seed(0))
draws = [torch.randn(B, 128, generator=g) for _ in range(6)]
got = traced_hip_step(solver, model, x, [t.clone() for t in draws])


def oracle(dtype):
    st = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    tr = Trainer("intro_tc", Net("conv", state=st, **C2), dataset_size=10000, beta_kl=0.5, beta_rec=0.75, beta_neg=512.0,
                 gamma_r=1e-8, clip=100.0, lr=2e-4)
    return traced_oracle_step(tr, x.to(dtype), [t.to(dtype) for t in draws])


o32, o64 = oracle(torch.float32), oracle(torch.float64)
for ph, part in enumerate(("encoder", "decoder")):
    ref = o64["grads"][ph]
    keys = [k for k in ref if k.startswith(part + ".")]
    scale = max(float(ref[k].abs().max()) for k in keys)
    rows = []
    for k in keys:
        eh = float((got["grads"][ph][k].double() - ref[k]).abs().max()) / scale
        eo = float((o32["grads"][ph][k].double() - ref[k]).abs().max()) / scale
        rows.append((eh, eo, float(ref[k].abs().max()) / scale, k))
    rows.sort(reverse=True)
    print(f"--- phase {'ED'[ph]} ({math}); errors vs fp64 oracle relative to the largest gradient tensor ({scale:.3e})")
    print("   HIP        oracle32   |tensor|   name")
    for eh, eo, mag, k in rows[:12]:
        print(f"   {eh:.2e}   {eo:.2e}   {mag:.2e}   {k}")
    print("   worst oracle32:", max(r[1] for r in rows))
for name in ("loss_enc", "loss_dec", "loss_kl", "loss_rec", "L2"):
    print(name, got["dict"][name], o32["dict"][name], o64["dict"][name])
