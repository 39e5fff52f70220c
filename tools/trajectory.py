"""Training trajectories of the four conv arithmetic modes on the benchmark configuration (c2, B=64): the same
weights, batches and device noise stream, STEPS intro-tc steps each; prints loss_rec / loss_kl / loss_enc / loss_dec
averaged over windows of 50 steps and the relative difference of every mode to exact fp32.  Chaotic divergence of
individual steps is expected (Adam's sign-like early updates amplify 1e-7 differences); the
window means show whether a mode trains differently; the yardstick is exact fp32 started from weights scaled by
(1 + 1e-7).   python3 tools/trajectory.py [STEPS] > gpurun_out/traj.json"""
import contextlib, io, json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "intro-tc-vae_amd"))
import models
from solvers.intro_tc import IntroTCSovler

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")


class _DS:
    def __len__(self):
        return 10000


def batches():
    """16 fixed batches of smooth synthetic images (random low-frequency fields), values in [0, 1]."""
    g = torch.Generator().manual_seed(42)
    low = torch.rand(16 * 64, 3, 8, 8, generator=g)
    img = torch.nn.functional.interpolate(low, size=64, mode="bilinear", align_corners=False)
    return [b.contiguous().to(dev) for b in img.view(16, 64, 3, 64, 64)]


def run(mode, perturb=False):
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.SoftIntroVAE(arch="conv", cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64).to(dev).train()
    if perturb:                                  # yardstick: exact fp32 from weights moved by one part in 10^7
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(1.0 + 1e-7)
    s = IntroTCSovler(_DS(), m, 64, torch.optim.Adam(m.encoder.parameters(), lr=2e-4),
                      torch.optim.Adam(m.decoder.parameters(), lr=2e-4), "mse", 0.5, 0.75, 512.0, 1e-8, dev,
                      mode != "fp32", None, clip=100.0)
    s.conv_math = mode
    s.enable_graph()
    torch.cuda.manual_seed(7)
    xs = batches()
    keys = ("loss_rec", "loss_kl", "loss_enc", "loss_dec")
    acc, out = {k: 0.0 for k in keys}, []
    for i in range(STEPS):
        d = s.train_step(xs[i % len(xs)], i)
        for k in keys:
            acc[k] += d[k]
        if (i + 1) % 50 == 0:
            out.append({k: acc[k] / 50 for k in keys})
            acc = {k: 0.0 for k in keys}
    return out


res = {m: run(m) for m in ("fp32", "f16x3", "bf16x6", "bf16x3")}
res["fp32_weights_x(1+1e-7)"] = run("fp32", perturb=True)
rel = {m: [{k: abs(w[k] - r[k]) / (abs(r[k]) + 1e-12) for k in w} for w, r in zip(res[m], res["fp32"])]
       for m in res if m != "fp32"}
print(json.dumps({"steps": STEPS, "window": 50, "config": "c2 conv 64x64x3 z=128 B=64, hipGraph, same seeds",
                  "window_means": res, "rel_diff_to_fp32": rel}, indent=1))
