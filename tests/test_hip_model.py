"""GPU parity tests of the model and the four solver steps against the golden vectors captured
from the unmodified reference (tests/golden/*.npz) and against the pinned CPU oracle.

Bar (north_star): losses / KL terms / reconstructions within 1e-4 relative of the fp32 CPU path
on identical seeds and inputs.  Gradients are compared at 1e-3 of each tensor's scale."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = dict(cdim=3, zdim=10, channels=(8, 16, 32), image_size=32)


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a, b):
    a, b = a.detach().double().cpu(), T(b).double() if not isinstance(b, torch.Tensor) else b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def load_state(npz, prefix):
    return {k[len(prefix):].replace("/", "."): T(npz[k]).clone() for k in npz.files if k.startswith(prefix)}


def build(arch, state):
    import models
    m = models.SoftIntroVAE(arch=arch, **TINY)
    m.load_state_dict(state, strict=True)
    return m.to(dev()).train()


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("arch", ["conv", "res", "inception"])
def test_model_forward_backward_golden(arch, fused):
    import ops
    g = np.load(os.path.join(GOLDEN, f"model_{arch}.npz"))
    model = build(arch, load_state(g, "init:")).set_fused(fused)
    x = T(g["x"]).to(dev())
    with ops.noise_queue([T(g["eps"])]):
        mu, logvar, z, rec = model(x)
    assert rel_err(mu, g["mu"]) < 1e-4 and rel_err(logvar, g["logvar"]) < 1e-4
    assert rel_err(z, g["z"]) < 1e-4 and rel_err(rec, g["rec"]) < 1e-4
    s = (rec * T(g["probe_img"]).to(dev())).sum() + (mu * T(g["probe_mu"]).to(dev())).sum() + (
        logvar * T(g["probe_lv"]).to(dev())).sum()
    assert abs(float(s.detach()) - float(g["scalar"])) < 1e-4 * abs(float(g["scalar"]))
    s.backward()
    n = 0
    for name, p in model.named_parameters():
        key = "grad:" + name.replace(".", "/")
        if key in g.files:
            assert rel_err(p.grad, g[key]) < 1e-3, name
            n += 1
        else:
            assert p.grad is None and "conv_expand" in name
    assert n > 10
    after = load_state(g, "after_train_fwd:")
    sd = model.state_dict()
    for k, v in after.items():
        if "running" in k:
            assert rel_err(sd[k], v) < 1e-4, k
        if "num_batches" in k:
            assert int(sd[k]) == int(v), k
    model.eval()
    with torch.no_grad():
        mu_e, lv_e = model.encode(x)
        rec_e = model.decode(mu_e)
    assert rel_err(mu_e, g["eval_mu"]) < 1e-4 and rel_err(rec_e, g["eval_rec"]) < 1e-4


@pytest.mark.parametrize("arch", ["conv", "res", "inception"])
@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
def test_forward_hooks_on_every_submodule(arch, math):
    """/root/reference/train.py:119-138 (anomaly detection): an `isnan` forward hook on every submodule.  With hooks
    registered the networks run the reference's module-by-module sequence: EVERY named submodule's hook fires, every
    output a hook sees is a completely written fp32 tensor (no NaN even when planes-only tensors are poisoned with NaN:
    HF._POISON), and mu / logvar / reconstruction equal the fused schedule's (golden values, 1e-4)."""
    import ops
    from hipvae import functional as HF
    g = np.load(os.path.join(GOLDEN, f"model_{arch}.npz"))
    model = build(arch, load_state(g, "init:"))
    x = T(g["x"]).to(dev())
    fired, bad = set(), []

    def make(name):
        def hook(mod, _, output):
            outs = output if isinstance(output, tuple) else [output]
            fired.add(name)
            for o in outs:
                if torch.isnan(o).any():
                    bad.append(name)
        return hook

    handles = [m.register_forward_hook(make(n)) for n, m in model.named_modules()]
    HF.set_conv_math(math)
    prev, HF._POISON[0] = HF._POISON[0], True
    try:
        with ops.noise_queue([T(g["eps"])]):
            mu, logvar, z, rec = model(x)
    finally:
        HF._POISON[0] = prev
        HF.set_conv_math("fp32")
    tol = 1e-4 if math == "fp32" else 2e-3
    assert rel_err(mu, g["mu"]) < tol and rel_err(logvar, g["logvar"]) < tol and rel_err(rec, g["rec"]) < tol
    assert not bad, bad
    # conv_expand of the plain conv block exists for the state dict only and is never called (models.py:49-54)
    expected = {n for n, m in model.named_modules() if not (arch == "conv" and n.endswith("conv_expand"))}
    assert fired == expected, sorted(expected - fired)
    (rec.sum() + mu.sum()).backward()           # the module-by-module graph is differentiable end to end
    assert all(p.grad is not None for n, p in model.named_parameters() if "conv_expand" not in n)
    for h in handles:
        h.remove()
    with ops.noise_queue([T(g["eps"])]):          # hooks gone: back on the fused schedule, same values
        mu2, _, _, rec2 = model(x)
    assert rel_err(mu2, g["mu"]) < 1e-4 and rel_err(rec2, g["rec"]) < 1e-4


def test_forward_hooks_at_planes_widths():
    """Channel widths that reach the planes kernels (bf16x3): on the fused schedule 8 of 13 conv inputs exist as planes
    only; with hooks registered (and those fp32 tensors poisoned) every hook still sees finite, written values and the
    outputs equal the fused schedule's to rounding."""
    import models
    import ops
    from hipvae import functional as HF
    torch.manual_seed(3)
    model = models.SoftIntroVAE(arch="conv", cdim=3, zdim=16, channels=(64, 64, 128), image_size=32).to(dev()).train()
    x = torch.rand(8, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(dev())
    eps = torch.randn(8, 16, generator=torch.Generator().manual_seed(2))
    HF.set_conv_math("bf16x3")
    prev, HF._POISON[0] = HF._POISON[0], True
    try:
        with ops.noise_queue([eps]):
            mu0, _, _, rec0 = model(x)
        bad, n = [], [0]

        def hook(mod, _, output):
            n[0] += 1
            if torch.isnan(output[0] if isinstance(output, tuple) else output).any():
                bad.append(type(mod).__name__)

        handles = [m.register_forward_hook(hook) for m in model.modules()]
        with ops.noise_queue([eps]):
            mu1, _, _, rec1 = model(x)
        for h in handles:
            h.remove()
    finally:
        HF._POISON[0] = prev
        HF.set_conv_math("fp32")
    assert not bad and n[0] > 60
    assert rel_err(mu1, mu0) < 1e-4 and rel_err(rec1, rec0) < 1e-4


class _DS:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


MATH_MODES = ["fp32", "bf16x6", "bf16x3", "f16x3"]


def make_solver(name, model, hp, loss_type="mse", math="fp32"):
    from solvers import IntroSolver, VAESolver
    from solvers.intro_tc import IntroTCSovler
    from solvers.tc import TCSovler
    cls = dict(vae=VAESolver, tc=TCSovler, intro=IntroSolver, intro_tc=IntroTCSovler)[name]
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=hp[5])
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=hp[5])
    kw = dict(dataset=_DS(int(hp[6])), model=model, batch_size=8, optimizer_e=opt_e, optimizer_d=opt_d,
              recon_loss_type=loss_type, beta_kl=hp[0], beta_rec=hp[1], device=dev(), use_amp=False,
              grad_scaler=None, writer=None, test_iter=1000, clip=hp[4])
    if name.startswith("intro"):
        kw.update(beta_neg=hp[2], gamma_r=hp[3])
    solver = cls(**kw)
    solver.conv_math = math
    return solver


def run_golden_steps(fname, arch, loss_type, names, nsteps):
    import ops
    g = np.load(os.path.join(GOLDEN, fname))
    hp = g["hp"]
    for name in names:
        model = build(arch, load_state(g, "init:"))
        solver = make_solver(name, model, hp, loss_type)
        kl_log, rec_log = [], []
        kl0, rec0 = solver.compute_kl_loss, solver.compute_rec_loss
        solver.compute_kl_loss = lambda *a, _f=kl0, **k: (kl_log.append(_f(*a, **k)), kl_log[-1])[1]
        solver.compute_rec_loss = lambda *a, _f=rec0, **k: (rec_log.append(_f(*a, **k)), rec_log[-1])[1]
        for s in range(nsteps):
            p = f"{name}:s{s}:"
            nd = len([k for k in g.files if k.startswith(p + "draw")])
            kl_log.clear(), rec_log.clear()
            with ops.noise_queue([T(g[p + f"draw{i}"]) for i in range(nd)]):
                d = solver.train_step(T(g[f"x{s}"]), s)
            got = np.array([d["loss_enc"], d["loss_dec"], d["loss_kl"], d["loss_rec"], d["L2"]])
            np.testing.assert_allclose(got, g[p + "dict"], rtol=1e-4 if s == 0 else 3e-4, err_msg=p)
            assert set(d) == {"loss_enc", "loss_dec", "loss_kl", "loss_rec", "L2"}
            assert all(isinstance(v, float) for v in d.values())
            # step 0 starts from identical weights: every hook output within 3e-4 of its scale.  Later
            # steps start from weights that already differ by O(lr) where Adam's first update
            # (+-lr * sign(g)) meets a near-zero gradient, and the per-sample beta_neg=512 TC terms
            # amplify that; they are held to 3e-3 (the returned losses above stay within 3e-4).
            tol = 3e-4 if s == 0 else 3e-3
            for i, t in enumerate(kl_log):
                ref = g[p + f"kl{i}"]
                assert float((t.detach().reshape(-1).cpu() - T(ref)).abs().max()) < tol * float(np.abs(ref).max()), (p, i)
            for i, t in enumerate(rec_log):
                assert rel_err(t.reshape(-1), g[p + f"rec{i}"]) < tol, (p, i)
        fin = load_state(g, f"{name}:final:")
        sd = model.state_dict()
        # Adam's first updates are +-lr*sign(g): where |g| is at rounding level the sign, and with it
        # 2*lr of that weight, can differ.  Bound the bulk tightly and every element by the Adam limit.
        diffs = torch.cat([(sd[k].detach().cpu() - v).abs().reshape(-1) for k, v in fin.items()
                           if v.dtype.is_floating_point and "running" not in k])
        assert float(diffs.max()) <= 2.05 * hp[5] * nsteps, (name, float(diffs.max()))
        assert float((diffs > 0.25 * hp[5] * nsteps).float().mean()) < 2e-3, name
        assert float(diffs.median()) < 0.02 * hp[5], name
        for k, v in fin.items():
            if "running" in k:
                assert rel_err(sd[k], v) < 1e-3, k
        # the reference leaves the encoder frozen / decoder trainable after an intro step
        if name.startswith("intro"):
            assert not next(model.encoder.parameters()).requires_grad
            assert next(model.decoder.parameters()).requires_grad


def test_steps_conv_golden():
    run_golden_steps("steps_conv.npz", "conv", "mse", ("vae", "tc", "intro", "intro_tc"), 2)


def test_use_amp_selects_split_fp16():
    """use_amp=True (the reference config's default, inert there) maps to the f16x3 conv arithmetic (fp32-class)."""
    import models
    from solvers.intro_tc import IntroTCSovler
    m = models.SoftIntroVAE(arch="conv", **TINY).to(dev())
    mk = lambda amp: IntroTCSovler(_DS(10), m, 4, torch.optim.Adam(m.encoder.parameters()),  # noqa: E731
                                   torch.optim.Adam(m.decoder.parameters()), "mse", 1.0, 1.0, 1.0, 1e-8, dev(), amp, None)
    assert mk(True).conv_math == "f16x3" and mk(False).conv_math == "fp32"


def test_steps_res_golden():
    run_golden_steps("steps_res.npz", "res", "mse", ("vae", "tc", "intro", "intro_tc"), 1)


def test_steps_bce_golden():
    run_golden_steps("steps_conv_bce.npz", "conv", "bce", ("vae", "intro_tc"), 1)


from step_trace import STEP_TOL, compare_traces, traced_hip_step, traced_oracle_step, worst  # noqa: E402


C2 = dict(cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)


@pytest.fixture(scope="module")
def c2_oracles():
    """One intro-TC step of the CPU oracle at the benchmark shape (B=8) in fp32 (the reference's precision) and in
    fp64 (ground truth for the quantities no fp32 evaluation pins to 1e-4), shared by the arithmetic modes."""
    import models
    from oracle.network import Net
    from oracle.steps import Trainer
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in models.SoftIntroVAE(arch="conv", **C2).state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    draws = [torch.randn(8, 128, generator=g) for _ in range(6)]
    out = dict(sd=sd, x=x, draws=draws)
    for name, dt in (("o32", torch.float32), ("o64", torch.float64)):
        st = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        tr = Trainer("intro_tc", Net("conv", state=st, **C2), dataset_size=10000, beta_kl=0.5, beta_rec=0.75,
                     beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4)
        out[name] = traced_oracle_step(tr, x.to(dt), [t.to(dt) for t in draws])
    return out


@pytest.mark.parametrize("math", MATH_MODES)
def test_intro_tc_step_64x64_vs_oracle(math, c2_oracles):
    """The benchmark shape at a batch the CPU oracle finishes in seconds (64x64x3, z=128,
    channels (64,128,256,512), B=8): one intro-TC step, HIP vs oracle on identical weights / draws,
    in every conv arithmetic (exact fp32 MFMA, bf16x6, bf16x3 = the benchmark's use_amp mode).
    Bar: every returned scalar within 1e-4 relative; every reconstruction / sample image, encoder output and
    KL / reconstruction hook output of phase E and the (mi, tc, dwkl) decomposition within 1e-4; per-tensor gradients
    of both phases and the phase-D tensors against the fp64 oracle within STEP_TOL[math]["ratio"] x the fp32 oracle's
    own error (tests/step_trace.py)."""
    import models
    from oracle import latent_math as lm
    model = models.SoftIntroVAE(arch="conv", **C2)
    model.load_state_dict(c2_oracles["sd"])
    model = model.to(dev()).train()
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 10000]
    solver = make_solver("intro_tc", model, hp, math=math)
    solver.batch_size = 8
    x, draws, ref = c2_oracles["x"], c2_oracles["draws"], c2_oracles["o32"]
    got = traced_hip_step(solver, model, x, [t.clone() for t in draws])
    d, r = got["dict"], ref["dict"]
    for k in ("loss_enc", "loss_dec", "loss_kl", "loss_rec", "L2"):
        # every loss / KL / reconstruction term within 1e-4 in every arithmetic.  L2 is the clipped
        # gradient-norm diagnostic: 2^-16-per-product rounding of the bf16x3 backward GEMMs shows there
        # first (measured 2.4e-4; fp32 and bf16x6 stay at 6e-5), so it is held to 1e-3 in that mode only.
        tol = 1e-3 if (k == "L2" and math == "bf16x3") else 1e-4
        assert abs(d[k] - r[k]) <= tol * abs(r[k]), (k, d[k], r[k])
    compare_traces(got, ref, c2_oracles["o64"], STEP_TOL[math], math)
    # decomposed KL terms (solvers/tc.py:104-121 as metrics) on the step's own posterior of the real batch
    mu_h, lv_h = got["encoded"][0]
    mu_o, lv_o = ref["encoded"][0]
    z_h, z_o = mu_h + draws[1] * (0.5 * lv_h).exp(), mu_o + draws[1] * (0.5 * lv_o).exp()
    dec_h = solver.kl_decomposition(*(t.to(dev()) for t in (z_h, mu_h, lv_h)))
    dec_o = lm.decomposition(z_o, mu_o, lv_o, 10000)
    e_dec = worst(zip(dec_h, dec_o))
    print(f"[{math}] (mi, tc, dwkl) decomposition {e_dec:.2e}")
    assert all(t.shape == (8,) for t in dec_h) and e_dec < STEP_TOL[math]["dec"]


@pytest.mark.parametrize("math,size,zdim,channels,B", [
    ("fp32", 128, 256, (64, 128, 256, 512, 512), 4), ("bf16x3", 128, 256, (64, 128, 256, 512, 512), 4),
    ("f16x3", 128, 256, (64, 128, 256, 512, 512), 4), ("bf16x3", 256, 512, (64, 128, 256, 512, 512, 512), 2),
    ("f16x3", 256, 512, (64, 128, 256, 512, 512, 512), 2)])
def test_intro_tc_step_large_images_vs_oracle(math, size, zdim, channels, B):
    """BASELINE configs[2] / configs[4] shapes (128x128x3, z=256; 256x256x3, z=512) at a small batch: layers wider
    than the band / transposing-read kernels take (W > 64) run on the 128-pixel-tile and in-kernel-split forms --
    same bar as the 64x64 test.  The gradient-norm diagnostic `L2` of the fp32-class modes is judged against the
    committed fp64 ground truth (tests/golden/large_images_fp64.json, made by make_large_fp64.py) with the fp32 oracle's
    own error as the yardstick: at 256x256 the fp32 oracle is 1.1e-4 off fp64 there (the HIP f16x3 step 1.2e-5)."""
    import models
    import ops
    from oracle.network import Net
    from oracle.steps import Trainer
    cfg = dict(cdim=3, zdim=zdim, channels=channels, image_size=size)
    torch.manual_seed(0)
    model = models.SoftIntroVAE(arch="conv", **cfg)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev()).train()
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 10000]
    solver = make_solver("intro_tc", model, hp, math=math)
    solver.batch_size = B
    g = torch.Generator().manual_seed(4321)
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(1))
    draws = [torch.randn(B, zdim, generator=g) for _ in range(6)]
    with ops.noise_queue(draws):
        d = solver.train_step(x, 0)
    tr = Trainer("intro_tc", Net("conv", state=sd, **cfg), dataset_size=10000, beta_kl=0.5, beta_rec=0.75,
                 beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4)
    ref = tr.step(x, draws)
    import json
    g64 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_images_fp64.json")))[str(size)]
    for k in ("loss_enc", "loss_dec", "loss_kl", "loss_rec", "L2"):
        if k != "L2":
            assert abs(ref[k] - g64[k]) <= 2e-6 * abs(g64[k]), ("fixture / oracle mismatch", k, ref[k], g64[k])
        if k == "L2" and math != "bf16x3":
            yard = abs(ref[k] - g64[k])
            assert abs(d[k] - g64[k]) <= 1.5 * yard + 1e-5 * abs(g64[k]), (k, d[k], ref[k], g64[k])
            continue
        tol = 1e-3 if k == "L2" else 1e-4
        assert abs(d[k] - ref[k]) <= tol * abs(ref[k]), (k, d[k], ref[k])


def test_graph_replay_equals_eager():
    """hipGraph mode (whole step captured once, replayed per step) gives the same trajectory as eager
    execution: same device RNG stream, same kernels, deterministic reductions."""
    import models
    cfg = dict(cdim=3, zdim=16, channels=(16, 32, 64), image_size=32)
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 1000]
    xs = [torch.rand(8, 3, 32, 32, generator=torch.Generator().manual_seed(i)).to(dev()) for i in range(7)]
    out = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        model = models.SoftIntroVAE(arch="conv", **cfg).to(dev()).train()
        solver = make_solver("intro_tc", model, hp)
        if mode == "graph":
            solver.enable_graph()
        torch.cuda.manual_seed(1234)
        out[mode] = [solver.train_step(x, i) for i, x in enumerate(xs)]
        if mode == "graph":
            assert solver._graph is not None, "graph was not captured"
            # eager code after replays sees the updated weights (packed-weight cache invalidated)
            with torch.no_grad():
                a = model.sample(torch.zeros(2, 16, device=dev()))
                model.set_fused(False)
                b = model.sample(torch.zeros(2, 16, device=dev()))
            assert rel_err(a, b) < 1e-6
        out[mode + "_w"] = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    for a, b in zip(out["eager"], out["graph"]):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * abs(a[k]) + 1e-9, (k, a[k], b[k])
    assert float((out["eager_w"] - out["graph_w"]).abs().max()) < 1e-6


# ---- solver x architecture x arithmetic at real channel widths (BASELINE configs[0]'s shape) -------------------------
C1 = dict(cdim=3, zdim=10, channels=(64, 128, 256), image_size=32)
_MATRIX_ORACLE = {}


def _matrix_oracle(name, arch):
    """One CPU-oracle step of solver ``name`` on architecture ``arch`` at the c1 shape (32x32x3, z=10, channels
    (64,128,256), B=16), cached per (solver, arch): initial weights, inputs, returned dict and hook outputs."""
    key = (name, arch)
    if key not in _MATRIX_ORACLE:
        import models
        from oracle.network import Net
        from oracle.steps import Trainer
        torch.manual_seed(5)
        sd = {k: v.clone() for k, v in models.SoftIntroVAE(arch=arch, **C1).state_dict().items()}
        g = torch.Generator().manual_seed(17)
        x = torch.rand(16, 3, 32, 32, generator=g)
        draws = [torch.randn(16, 10, generator=g) for _ in range(6 if name.startswith("intro") else 1)]
        tr = Trainer(name, Net(arch, state={k: v.clone() for k, v in sd.items()}, **C1), dataset_size=10000, beta_kl=0.5,
                     beta_rec=0.75, beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4)
        ref = tr.step(x, draws)
        _MATRIX_ORACLE[key] = dict(sd=sd, x=x, draws=draws, ref=ref, kl=[t.clone() for t in tr.trace["kl"]],
                                   rec=[t.clone() for t in tr.trace["rec"]])
    return _MATRIX_ORACLE[key]


@pytest.mark.parametrize("math", ["fp32", "f16x3", "bf16x6", "bf16x3"])
@pytest.mark.parametrize("arch", ["conv", "res", "inception"])
@pytest.mark.parametrize("name", ["vae", "tc", "intro", "intro_tc"])
def test_solver_arch_math_matrix_vs_oracle(name, arch, math):
    """Every solver (solvers/vae.py:89-136, tc.py:58-89, intro.py:56-196, intro_tc.py) on every architecture
    (/root/reference/models.py:8-182) in every conv arithmetic, at channel widths that reach the planes / band / matrix-
    core kernels, against the CPU oracle on identical weights, inputs and draws: the returned dict within 1e-4 (the clip
    norm within 1e-3 in bf16x3), every compute_kl_loss / compute_rec_loss hook output of the phase that starts from the
    identical weights within 3e-4 of its scale, the hook outputs behind the encoder's Adam update (phase D of the intro
    solvers: weights whose gradient sits at rounding level move 2*lr apart) within 3e-3."""
    import models
    import ops
    o = _matrix_oracle(name, arch)
    model = models.SoftIntroVAE(arch=arch, **C1)
    model.load_state_dict(o["sd"])
    model = model.to(dev()).train()
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 10000]
    solver = make_solver(name, model, hp, math=math)
    solver.batch_size = 16
    kl_log, rec_log = [], []
    kl0, rec0 = solver.compute_kl_loss, solver.compute_rec_loss
    solver.compute_kl_loss = lambda *a, _f=kl0, **k: (kl_log.append(_f(*a, **k)), kl_log[-1])[1]
    solver.compute_rec_loss = lambda *a, _f=rec0, **k: (rec_log.append(_f(*a, **k)), rec_log[-1])[1]
    with ops.noise_queue([t.clone() for t in o["draws"]]):
        d = solver.train_step(o["x"], 0)
    for k in ("loss_enc", "loss_dec", "loss_kl", "loss_rec", "L2"):
        tol = 1e-3 if (k == "L2" and math == "bf16x3") else 1e-4
        assert abs(d[k] - o["ref"][k]) <= tol * abs(o["ref"][k]), (k, d[k], o["ref"][k])
    assert len(kl_log) == len(o["kl"]) and len(rec_log) == len(o["rec"])
    first = 3 if name.startswith("intro") else 99          # hook calls of the phase that starts from identical weights
    for logs, refs in ((kl_log, o["kl"]), (rec_log, o["rec"])):
        for i, (t, r) in enumerate(zip(logs, refs)):
            tol = 3e-4 if i < first else 3e-3
            err = float((t.detach().reshape(-1).cpu() - r).abs().max()) / float(r.abs().max())
            assert err < tol, (i, err)


FULL_SIZE = {  # BASELINE configs[1] / [2] / [4] at their full per-GPU batch
    "c2": (C2, 64),
    "c3": (dict(cdim=3, zdim=256, channels=(64, 128, 256, 512, 512), image_size=128), 128),
    "c5": (dict(cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256), 32),
}


@pytest.mark.parametrize("cfg,math", [("c2", "f16x3"), ("c2", "bf16x3"), ("c3", "f16x3"), ("c5", "f16x3")])
def test_full_size_step_properties(cfg, math):
    """BASELINE configs[1] (64x64x3, z=128, batch 64 -- the bench line's workload), configs[2] (128x128x3, z=256, batch
    128) and configs[4] (256x256x3, z=512, 32 per GPU) at their FULL size, where the CPU oracle takes minutes per step:
    size-independent properties instead.  (a) The same steps issued as 7 batched passes (persistent band kernels, no
    split-K) and as 13 passes one by one (one-tile kernels, split-K on the small layers) agree -- two different sets of
    launch shapes for every conv, BatchNorm and weight-gradient layer; (b) hipGraph replays reproduce the eager
    trajectory; (c) every returned scalar is finite and the reconstruction loss falls over the steps."""
    import models
    net, B = FULL_SIZE[cfg]
    S = net["image_size"]
    torch.manual_seed(11)
    init = models.SoftIntroVAE(arch="conv", **net).state_dict()
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 10000]
    g = torch.Generator().manual_seed(21)
    nx = 6 if cfg == "c2" else 3
    xs = [torch.rand(B, 3, S, S, generator=g).to(dev()) for _ in range(nx)]

    def run(batched, graph, nsteps):
        model = models.SoftIntroVAE(arch="conv", **net)
        model.load_state_dict(init)
        model = model.to(dev()).train()
        solver = make_solver("intro_tc", model, hp, math=math)
        solver.batch_size = B
        solver.batch_passes = batched
        if graph:
            solver.enable_graph()
        torch.cuda.manual_seed(77)                       # the step draws its noise from the device generator
        res = [solver.train_step(xs[i % nx], i) for i in range(nsteps)]
        if graph:
            assert solver._graph is not None
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        del solver, model
        torch.cuda.empty_cache()
        return res, w

    (rb, wb), (ru, wu) = run(True, False, 3), run(False, False, 3)
    # step 0 (identical weights): every scalar to 1e-5 (measured 3e-7 / 1e-6, the clip norm L2 1.5e-5 -> 1e-4).  From
    # step 1 on the two runs are different trajectories: Adam turns the sign of a near-zero gradient into +-lr, and the
    # exp(-2*scale*(rec + 512*kl)) terms of loss_enc amplify that (measured 8e-4 / 2e-3 at steps 1 / 2, 1.2e-3 / 5e-3 in
    # exact fp32 mode as well): held to 2e-2 there, and the weights through the size of the update (3 steps x 2e-4)
    for i, (a, b) in enumerate(zip(rb, ru)):
        for k in a:
            tol = (1e-4 if k == "L2" else 1e-5) if i == 0 else 2e-2
            assert a[k] == a[k] and abs(a[k] - b[k]) <= tol * abs(b[k]) + 1e-7, (i, k, a[k], b[k])
    assert float((wb - wu).abs().max()) <= 2.05 * 3 * 2e-4
    assert float((wb - wu).abs().mean()) < 1.5e-4       # a quarter of the largest possible total update
    assert rb[2]["loss_rec"] < rb[0]["loss_rec"]
    ng = 6 if cfg == "c2" else 5
    (rg, wg), (re_, we) = run(True, True, ng), run(True, False, ng)
    for a, b in zip(rg, re_):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-9, (k, a[k], b[k])
    assert float((wg - we).abs().max()) < 1e-6


class StubWriter:
    """Stands in for torch.utils.tensorboard.SummaryWriter: records every call the solvers make."""

    def __init__(self):
        self.calls = []

    def add_images(self, tag, img_tensor, global_step=None):
        assert img_tensor.device.type == "cpu"          # the reference hands over ``.data.cpu()`` (vae.py:151-160)
        self.calls.append(("add_images", tag, img_tensor.clone(), global_step))

    def add_scalar(self, tag, value, global_step=None):
        self.calls.append(("add_scalar", tag, float(value), global_step))

    def add_scalars(self, tag, values, global_step=None):
        self.calls.append(("add_scalars", tag, {k: float(v) for k, v in values.items()}, global_step))

    def flush(self):
        self.calls.append(("flush",))

    def of(self, kind, tag):
        return [c for c in self.calls if c[0] == kind and c[1] == tag]


@pytest.mark.parametrize("name", ["intro_tc", "vae"])
def test_image_logging_grid_vs_oracle(name):
    """with a writer set and cur_iter % test_iter == 0 the step logs the [real | deterministic
    reconstruction | fake] grid (solvers/vae.py:138-163, intro.py:187-191): the deterministic reconstruction is
    ``model(batch, deterministic=True)`` on the UPDATED weights with BatchNorm still in train mode (batch statistics,
    running buffers advanced once more); ``fake`` is phase D's sample for the intro solvers and a fresh N(0,1)
    sample for the VAE solvers.  Checked against the oracle, with the scalars the reference logs beside it."""
    import ops
    from oracle.network import Net
    from oracle.steps import Trainer
    from utils import SingletonWriter
    g = np.load(os.path.join(GOLDEN, "steps_conv.npz"))
    hp = g["hp"]
    state = load_state(g, "init:")
    model = build("conv", state)
    solver = make_solver(name, model, hp)
    w = StubWriter()
    solver.writer, solver.test_iter = w, 5
    SingletonWriter().writer, SingletonWriter().cur_iter, SingletonWriter().test_iter = w, 10, 5
    try:
        x = T(g["x0"])
        p = f"{name}:s0:"
        nd = len([k for k in g.files if k.startswith(p + "draw")])
        draws = [T(g[p + f"draw{i}"]) for i in range(nd)]
        torch.cuda.manual_seed(77)
        with ops.noise_queue([t.clone() for t in draws]):
            d = solver.train_step(x, 10)
    finally:
        SingletonWriter().writer = None
    np.testing.assert_allclose([d["loss_enc"], d["loss_dec"], d["loss_kl"], d["loss_rec"], d["L2"]], g[p + "dict"], rtol=1e-4)
    tr = Trainer(name, Net("conv", state={k: v.clone() for k, v in state.items()}, **TINY), dataset_size=int(hp[6]),
                 beta_kl=hp[0], beta_rec=hp[1], beta_neg=hp[2], gamma_r=hp[3], clip=hp[4], lr=hp[5])
    ref = tr.step(x, draws)
    if name == "vae":
        torch.cuda.manual_seed(77)
        noise = torch.randn(size=(x.size(0), TINY["zdim"]), device=dev()).cpu()    # the helper's draw (vae.py:141-143)
        with torch.no_grad():
            fake_ref = tr.net.decode(noise)
    else:
        fake_ref = tr.trace["last_fake"]
    with torch.no_grad():
        rec_det = tr.net.decode(tr.net.encode(x)[0])
    grid_ref = torch.cat([x, rec_det, fake_ref], dim=0)
    (call,) = w.of("add_images", "reconstructions")
    assert call[3] == 10 and call[2].shape == grid_ref.shape == (3 * x.size(0), 3, 32, 32)
    assert rel_err(call[2], grid_ref) < 1e-4
    # BatchNorm buffers after the logging pass (one more train-mode forward of both halves) equal the oracle's
    sd = model.state_dict()
    for k, v in tr.net.sd.items():
        if "running" in k:
            assert rel_err(sd[k], v) < 1e-3, k
        if "num_batches" in k:
            assert int(sd[k]) == int(v), k
    # scalars beside the grid
    (losses,) = w.of("add_scalars", "losses")
    assert abs(losses[2]["r_loss"] - ref["loss_rec"]) <= 1e-4 * abs(ref["loss_rec"])
    assert abs(losses[2]["kl_loss"] - ref["loss_kl"]) <= 1e-4 * abs(ref["loss_kl"])
    assert w.of("add_scalar", "fc_grad_norm") and w.calls[-1] == ("flush",)
    if name == "intro_tc":
        (norms,) = w.of("add_scalars", "total_norm")
        np.testing.assert_allclose([norms[2]["E"], norms[2]["D"]], tr.trace["norms"], rtol=1e-4)
        assert abs(w.of("add_scalar", "lossE")[0][2] - ref["loss_enc"]) <= 1e-4 * abs(ref["loss_enc"])
        assert abs(w.of("add_scalar", "lossD")[0][2] - ref["loss_dec"]) <= 1e-4 * abs(ref["loss_dec"])
        assert abs(losses[2]["expelbo_f"] - tr.trace["expelbo"][1]) <= 1e-4 * abs(tr.trace["expelbo"][1])
        assert w.of("add_scalar", "diff_kl")
    # off-iteration: nothing but scalars
    w.calls.clear()
    with ops.noise_queue([t.clone() for t in draws]):
        solver.train_step(x, 11)
    assert not w.of("add_images", "reconstructions")


@pytest.mark.parametrize("graph", [False, True])
def test_prefetch_loader_equals_direct_feeding(graph):
    """N steps fed by hipvae.loader.PrefetchLoader (pinned staging + H2D on a side stream, WrappedDataLoader
    surface of dataset.py:16-27 / train.py:146-159) from a synthetic Dataset equal the same steps fed with the same
    tensors directly -- eagerly and with the step captured as a hipGraph (the copy lands while the previous replay runs)."""
    import models
    from torch.utils.data import DataLoader, TensorDataset
    from hipvae.loader import PrefetchLoader
    cfg = dict(cdim=3, zdim=16, channels=(16, 32, 64), image_size=32)
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 1000]
    g = torch.Generator().manual_seed(3)
    images, labels = torch.rand(60, 3, 32, 32, generator=g), torch.arange(60)
    dl = DataLoader(TensorDataset(images, labels), batch_size=8, shuffle=False)      # 7 full batches + one of 4
    seen = []

    def batch_to_device(x, y):                # train.py:152-156
        assert x.is_cuda and y.is_cuda and x.max() <= 1.0 and x.min() >= 0.0
        seen.append(y.cpu())
        return x.to(dev()), y.to(dev())

    out = {}
    for mode in ("direct", "loader"):
        torch.manual_seed(0)
        model = models.SoftIntroVAE(arch="conv", **cfg).to(dev()).train()
        solver = make_solver("intro_tc", model, hp)
        if graph:
            solver.enable_graph()
        torch.cuda.manual_seed(99)
        res = []
        if mode == "direct":
            for i, (x, _) in enumerate(dl):
                res.append(solver.train_step(x.to(dev()), i))
        else:
            loader = PrefetchLoader(dl, batch_to_device)
            assert len(loader) == len(dl) == 8
            for epoch in range(1):
                for i, batch in enumerate(loader):
                    res.append(solver.train_step(batch[0], i))
        out[mode] = res
        out[mode + "_w"] = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    assert len(out["direct"]) == len(out["loader"]) == 8
    assert torch.equal(torch.cat(seen), labels)               # every sample once, in order
    for a, b in zip(out["direct"], out["loader"]):
        for k in a:
            assert a[k] == b[k], (k, a[k], b[k])              # same kernels on the same bytes: bit-identical
    assert torch.equal(out["direct_w"], out["loader_w"])
    # a second epoch over the same loader object reuses the pinned / device rings
    loader = PrefetchLoader(dl, None, depth=2)
    got = torch.cat([b[0].cpu() for b in loader] + [b[0].cpu() for b in loader])
    assert torch.equal(got, torch.cat([images, images]))


def test_prefetch_loader_flips_on_the_gpu_and_survives_an_abandoned_epoch():
    """SURVEY 8(f3), second half: transforms.RandomHorizontalFlip(p=0.5) (/root/reference/dataset.py:219-224) applied per
    sample on the device behind the H2D copy: every delivered image is its source or the exact mirror of it, about half
    are mirrored, labels pass untouched, the coin flips are reproducible from ``flip_seed``; an epoch abandoned by
    ``break`` leaves no producer behind and the next epoch delivers every sample once, in order."""
    from torch.utils.data import DataLoader, TensorDataset
    from hipvae.loader import PrefetchLoader
    g = torch.Generator().manual_seed(5)
    images, labels = torch.rand(200, 3, 16, 32, generator=g), torch.arange(200)
    dl = DataLoader(TensorDataset(images, labels), batch_size=25, shuffle=False)

    def run(seed):
        xs, ys = [], []
        for x, y in PrefetchLoader(dl, None, flip_p=0.5, flip_seed=seed):
            xs.append(x.cpu()), ys.append(y.cpu())
        return torch.cat(xs), torch.cat(ys)

    x1, y1 = run(7)
    assert torch.equal(y1, labels)
    same = (x1 == images).flatten(1).all(1)
    mirrored = (x1 == images.flip(3)).flatten(1).all(1)
    assert bool((same | mirrored).all()) and 60 < int(mirrored.sum()) < 140
    x2, _ = run(7)
    x3, _ = run(8)
    assert torch.equal(x1, x2) and not torch.equal(x1, x3)
    # abandon an epoch early, then iterate again
    loader = PrefetchLoader(dl, None, depth=2)
    for i, (x, y) in enumerate(loader):
        if i == 2:
            break
    assert not loader._thread.is_alive()
    got = torch.cat([b[1].cpu() for b in loader])
    assert torch.equal(got, labels)


@pytest.mark.parametrize("arch", ["conv", "res", "inception"])
def test_bn_groups_model_pass_equals_separate_passes(arch):
    """models.bn_groups(2): one forward of two stacked batches == two forwards (outputs, running buffers, parameter
    gradients), for every block architecture, fused and layer-by-layer schedules."""
    import models
    g = np.load(os.path.join(GOLDEN, f"model_{arch}.npz"))
    gen = torch.Generator().manual_seed(5)
    xa, xb = torch.rand(4, 3, 32, 32, generator=gen).to(dev()), torch.rand(4, 3, 32, 32, generator=gen).to(dev())
    probe = torch.randn(8, 3, 32, 32, generator=gen).to(dev())
    for fused in (True, False):
        res = {}
        for mode in ("stacked", "separate"):
            model = build(arch, load_state(g, "init:")).set_fused(fused)
            if mode == "stacked":
                with models.bn_groups(2):
                    mu, lv = model.encode(torch.cat([xa, xb]))
                    rec = model.decode(mu)
            else:
                outs = [model.encode(t) for t in (xa, xb)]
                mu, lv = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
                rec = torch.cat([model.decode(o[0]) for o in outs])
            ((rec * probe).sum() + lv.sum()).backward()
            res[mode] = (mu.detach(), rec.detach(), {k: v.clone() for k, v in model.state_dict().items()},
                         {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        s, r = res["stacked"], res["separate"]
        assert rel_err(s[0], r[0]) < 1e-6 and rel_err(s[1], r[1]) < 1e-6
        for k, v in r[2].items():
            if "running" in k or "num_batches" in k:
                assert rel_err(s[2][k].float(), v.float()) < 1e-6, k
        assert s[3].keys() == r[3].keys()
        for k, v in r[3].items():
            assert rel_err(s[3][k], v) < 2e-5, k


def test_batched_passes_equal_unbatched_step():
    """IntroTCSovler with batch_passes (7 batched passes) == the 13 passes issued one by one: two steps, every returned
    scalar and the final weights (the weight-gradient sums associate differently: 1e-5, weights 1e-6)."""
    import ops
    g = np.load(os.path.join(GOLDEN, "steps_conv.npz"))
    hp = g["hp"]
    out = {}
    for batched in (True, False):
        model = build("conv", load_state(g, "init:"))
        solver = make_solver("intro_tc", model, hp)
        solver.batch_passes = batched
        res = []
        for s in range(2):
            p = f"intro_tc:s{s}:"
            with ops.noise_queue([T(g[p + f"draw{i}"]) for i in range(6)]):
                res.append(solver.train_step(T(g[f"x{s}"]), s))
        out[batched] = (res, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu(),
                        {k: v.clone().cpu() for k, v in model.state_dict().items() if "running" in k})
    for a, b in zip(out[True][0], out[False][0]):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]), (k, a[k], b[k])
    assert float((out[True][1] - out[False][1]).abs().max()) < 2.05 * 2e-4 * 2
    assert float(((out[True][1] - out[False][1]).abs() > 1e-6).float().mean()) < 1e-3
    for k, v in out[False][2].items():
        assert rel_err(out[True][2][k], v) < 1e-5, k


def test_group_weight_packing_equals_per_layer_packing():
    """After an optimiser step the split-bf16 conv operands of a whole network are re-packed by one launch per
    direction (device-resident descriptor table): three intro-TC steps at the benchmark shape are bit-identical to
    packing layer by layer, and the batched path was really taken."""
    import models
    import ops
    from hipvae import functional as HF
    torch.manual_seed(3)
    init = models.SoftIntroVAE(arch="res", **C2).state_dict()
    hp = [0.5, 0.75, 512.0, 1e-8, 100.0, 2e-4, 10000]
    g = torch.Generator().manual_seed(5)
    xs = [torch.rand(8, 3, 64, 64, generator=g).to(dev()) for _ in range(3)]
    draws = [[torch.randn(8, 128, generator=g).to(dev()) for _ in range(6)] for _ in range(3)]
    out, tables = {}, 0
    for batched in (True, False):
        HF._PACK_BATCH[0] = batched
        try:
            model = models.SoftIntroVAE(arch="res", **C2)
            model.load_state_dict(init)
            model = model.to(dev()).train()
            solver = make_solver("intro_tc", model, hp, math="bf16x3")
            solver.batch_size = 8
            res = []
            for s in range(3):
                with ops.noise_queue([t.clone() for t in draws[s]]):
                    res.append(solver.train_step(xs[s], s))
            if batched:
                grp = {id(v): v for v in (HF._PACK_GROUPS.get(id(p)) for p in model.parameters()) if v is not None}
                assert len(grp) == 2                                    # encoder and decoder
                tables = sum(len(v.tables) for v in grp.values())
                assert all(t[1] > 4 for v in grp.values() for t in v.tables.values())
        finally:
            HF._PACK_BATCH[0] = True
        out[batched] = (res, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
    assert tables >= 3                                                  # forward + data-gradient tables were built
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1])


def test_adam_state_roundtrip_and_ownership_carryover():
    """The fused Adam's moments live in flat buffers; they are mirrored into ``optimizer.state`` (so
    ``optimizer.state_dict()`` / ``load_state_dict()`` round-trip) and carried over when the parameters are moved
    out of the flat buffer (``model.float()`` / ``.to()`` / a manual re-point after the first step)."""
    import ops
    g = np.load(os.path.join(GOLDEN, "steps_conv.npz"))
    hp = g["hp"]
    draws = [[T(g[f"intro_tc:s{s}:draw{i}"]) for i in range(6)] for s in range(2)]
    xs = [T(g["x0"]), T(g["x1"])]

    def steps(solver, which):
        out = []
        for s in which:
            with ops.noise_queue([t.clone() for t in draws[s]]):
                out.append(solver.train_step(xs[s], s))
        return out

    # reference trajectory: step 0 then step 1, uninterrupted
    m0 = build("conv", load_state(g, "init:"))
    s0 = make_solver("intro_tc", m0, hp)
    ref = steps(s0, (0, 1))
    # (a) state_dict round trip after step 0 into a fresh model / optimizers
    m1 = build("conv", load_state(g, "init:"))
    s1 = make_solver("intro_tc", m1, hp)
    first = steps(s1, (0,))
    sd_e, sd_d, sd_m = s1.optimizer_e.state_dict(), s1.optimizer_d.state_dict(), m1.state_dict()
    assert len(sd_e["state"]) == len(list(m1.encoder.parameters())) and int(sd_e["state"][0]["step"]) == 1
    assert float(sd_e["state"][0]["exp_avg"].abs().max()) > 0
    m2 = build("conv", {k: v.cpu() for k, v in sd_m.items()})
    s2 = make_solver("intro_tc", m2, hp)
    s2.optimizer_e.load_state_dict(sd_e)
    s2.optimizer_d.load_state_dict(sd_d)
    second = steps(s2, (1,))
    assert first[0] == ref[0]
    # L2 is left out: the clip norm runs over every parameter with a gradient (intro.py:113-115), and in the resumed run
    # the frozen half's STALE gradients of the previous step are gone (the reference's checkpoints do not hold .grad either)
    for k in ("loss_enc", "loss_dec", "loss_kl", "loss_rec"):
        assert abs(second[0][k] - ref[1][k]) <= 1e-6 * abs(ref[1][k]), (k, second[0][k], ref[1][k])
    # (b) ownership lost after step 0: parameters re-pointed out of the flat buffers
    m3 = build("conv", load_state(g, "init:"))
    s3 = make_solver("intro_tc", m3, hp)
    steps(s3, (0,))
    for p in m3.parameters():
        p.data = p.data.clone()
    third = steps(s3, (1,))
    for k in ref[1]:
        assert abs(third[0][k] - ref[1][k]) <= 1e-6 * abs(ref[1][k]), (k, third[0][k], ref[1][k])
    wa = torch.cat([p.detach().reshape(-1) for p in m0.parameters()])
    wb = torch.cat([p.detach().reshape(-1) for p in m3.parameters()])
    assert float((wa - wb).abs().max()) < 1e-7
