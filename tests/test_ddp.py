"""Data-parallel path (hipvae.ddp): world_size-2 runs.

CPU (gloo): the collective algebra -- all-gather of mu with its reduce-scatter adjoint, gradient
averaging, global importance-weight indexing -- reproduces the full-batch estimator and its
gradients (checked with the pinned oracle's formulas).
GPU (gloo over two processes sharing cuda:0; NCCL needs one device per rank): a complete
IntroTCSovler step sharded 2 x B/2 with Sync-BN reproduces the reference's single-process
golden step (tests/golden/steps_conv.npz)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, world, port, backend="gloo"):
    for p in (os.path.join(ROOT, "intro-tc-vae_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


def _cpu_worker(rank, world, port, out):
    _setup(rank, world, port)
    from hipvae import ddp
    from oracle import latent_math as lm
    ddp.init(sync_bn=False)
    torch.manual_seed(0)
    B, D, N = 12, 7, 500
    mu = torch.randn(B, D)
    lv = -2 + torch.randn(B, D)
    z = mu + torch.randn(B, D) * (0.5 * lv).exp()
    Bl = B // world
    sl = slice(rank * Bl, (rank + 1) * Bl)

    def tc_rows(zr, mu_all, lvr, off):
        # ops.py:80-84,104-115 restricted to rows [off, off+len) of the global batch
        lp = lm.log_density_clamped_var(zr.unsqueeze(1), mu_all.unsqueeze(0), lvr.unsqueeze(1))
        lw = lm.log_importance_weights(mu_all.shape[0], N)[off:off + zr.shape[0]]
        return torch.logsumexp(lw + lp.sum(2), 1) - torch.logsumexp(lw.unsqueeze(2) + lp, 1).sum(1)

    # full-batch reference on every rank
    zf, mf, lf = (t.clone().requires_grad_(True) for t in (z, mu, lv))
    full = lm.total_correlation(zf, mf, lf, N, "none")
    full.mean().backward()
    # sharded: local rows, gathered means, mean over local rows, gradients averaged over ranks
    zl, ml, ll = (t[sl].clone().requires_grad_(True) for t in (z, mu, lv))
    mu_all = ddp.all_gather_rows(ml)
    assert mu_all.shape == (B, D) and ddp.row_offset(Bl) == rank * Bl
    loc = tc_rows(zl, mu_all, ll, ddp.row_offset(Bl))
    loc.mean().backward()
    ok = torch.allclose(loc, full[sl].detach(), rtol=1e-5, atol=1e-5)
    # d(global mean)/d(local leaf) = (1/world) * d(sum_r local mean_r)/d leaf: what grad averaging of
    # the downstream parameters applies; here the leaves themselves are compared
    for got, ref in ((zl.grad, zf.grad[sl]), (ml.grad, mf.grad[sl]), (ll.grad, lf.grad[sl])):
        ok = ok and torch.allclose(got / world, ref, rtol=1e-4, atol=1e-6)
    flat = torch.full((5,), float(rank + 1))
    ddp.average_(flat)
    ok = ok and torch.allclose(flat, torch.full((5,), (1 + world) / 2))
    v = torch.tensor([float(rank)])
    ddp.mean_scalars_(v)
    ok = ok and abs(float(v) - (world - 1) / 2) < 1e-6
    out[rank] = bool(ok)
    ddp.shutdown()
    dist.destroy_process_group()


def test_ddp_collective_algebra_cpu():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_cpu_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _gpu_worker(rank, world, port, out, backend="gloo"):
    _setup(rank, world, port, backend)
    import models
    import ops
    from hipvae import ddp
    from solvers.intro_tc import IntroTCSovler
    # nccl on the one GPU of the box: a group of ONE rank with the data-parallel paths forced on, so that every
    # collective of the step (async AVG all-reduce + wait, all_gather_into_tensor / reduce_scatter_tensor, Sync-BN
    # moments) goes through RCCL on the device
    ctx = ddp.init(sync_bn=True, force=(backend == "nccl"))
    assert ddp.get() is ctx
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLDEN, "steps_conv.npz"))
    hp = g["hp"]
    state = {k[5:].replace("/", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("init:")}
    model = models.SoftIntroVAE(arch="conv", cdim=3, zdim=10, channels=(8, 16, 32), image_size=32)
    model.load_state_dict(state)
    model = model.to(dev).train()

    class DS:
        def __len__(self):
            return int(hp[6])

    B = 8
    Bl = B // dist.get_world_size()
    sl = slice(rank * Bl, (rank + 1) * Bl)
    solver = IntroTCSovler(DS(), model, Bl, torch.optim.Adam(model.encoder.parameters(), lr=hp[5]),
                           torch.optim.Adam(model.decoder.parameters(), lr=hp[5]), "mse", hp[0], hp[1], hp[2], hp[3],
                           dev, False, None, clip=hp[4])
    res = []
    for s in range(2):
        p = f"intro_tc:s{s}:"
        draws = [torch.from_numpy(g[p + f"draw{i}"])[sl] for i in range(6)]
        with ops.noise_queue(draws):
            d = solver.train_step(torch.from_numpy(g[f"x{s}"])[sl], s)
        res.append([d["loss_enc"], d["loss_dec"], d["loss_kl"], d["loss_rec"], d["L2"]])
    fin = {k[len("intro_tc:final:"):].replace("/", "."): torch.from_numpy(g[k]) for k in g.files
           if k.startswith("intro_tc:final:")}
    sd = model.state_dict()
    diffs = torch.cat([(sd[k].detach().cpu() - v).abs().reshape(-1) for k, v in fin.items()
                       if v.dtype.is_floating_point and "running" not in k])
    run = max(float((sd[k].cpu() - v).abs().max() / (v.abs().max() + 1e-12)) for k, v in fin.items() if "running" in k)
    out[rank] = dict(res=res, max=float(diffs.max()), frac=float((diffs > 0.25 * hp[5] * 2).float().mean()), run=run)
    ddp.shutdown()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_step_matches_single_process_golden():
    world, port = 2, _free_port()
    g = np.load(os.path.join(GOLDEN, "steps_conv.npz"))
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_gpu_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r in (0, 1):
        for s in range(2):
            np.testing.assert_allclose(res[r]["res"][s], g[f"intro_tc:s{s}:dict"], rtol=1e-4 if s == 0 else 1e-3)
        assert res[r]["max"] <= 2.05 * 2e-4 * 2 and res[r]["frac"] < 5e-3 and res[r]["run"] < 1e-3
    assert res[0]["res"] == res[1]["res"]          # every rank reports the global scalars


@pytest.mark.gpu
def test_ddp_step_through_rccl_single_rank():
    """backend "nccl" (= RCCL) on the one device of the test box: the step's collectives run on the GPU through the
    production (non-gloo) branches of hipvae.ddp and must leave the golden single-process step unchanged."""
    port = _free_port()
    g = np.load(os.path.join(GOLDEN, "steps_conv.npz"))
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_gpu_worker, args=(1, port, out, "nccl"), nprocs=1, join=True)
        res = dict(out)
    for s in range(2):
        np.testing.assert_allclose(res[0]["res"][s], g[f"intro_tc:s{s}:dict"], rtol=1e-4 if s == 0 else 1e-3)
    assert res[0]["max"] <= 2.05 * 2e-4 * 2 and res[0]["frac"] < 5e-3 and res[0]["run"] < 1e-3


def _graph_worker(rank, world, port, out):
    os.environ["ITCV_DDP_GRAPH"] = "1"
    _setup(rank, world, port, "nccl")
    import models
    from hipvae import ddp
    from solvers.intro_tc import IntroTCSovler
    dev = torch.device("cuda:0")

    class DS:
        def __len__(self):
            return 1000

    def run(graph, forced):
        ddp.init(sync_bn=True, force=forced)
        torch.manual_seed(3)
        model = models.SoftIntroVAE(arch="conv", cdim=3, zdim=10, channels=(8, 16, 32), image_size=32).to(dev).train()
        solver = IntroTCSovler(DS(), model, 8, torch.optim.Adam(model.encoder.parameters(), lr=2e-4),
                               torch.optim.Adam(model.decoder.parameters(), lr=2e-4), "mse", 1.0, 1.0, 256.0, 1e-8,
                               dev, False, None, clip=100.0)
        if graph:
            solver.enable_graph()
        torch.manual_seed(5)
        xs = [torch.rand(8, 3, 32, 32).to(dev) for _ in range(6)]
        res = [solver.train_step(x, i) for i, x in enumerate(xs)]
        captured = getattr(solver, "_graph", None) is not None
        ddp.shutdown()
        return res, captured, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()

    eager, cap_e, p_e = run(False, True)
    graph, cap_g, p_g = run(True, True)
    out[rank] = dict(cap_e=cap_e, cap_g=cap_g, eager=[list(d.values()) for d in eager],
                     graph=[list(d.values()) for d in graph], pdiff=float((p_e - p_g).abs().max()))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_step_graph_capture_with_rccl():
    """ITCV_DDP_GRAPH=1: the data-parallel step (RCCL all-reduces, all-gather / reduce-scatter, Sync-BN moments, the
    deferred gradient average) captured into the step hipGraph and replayed equals the eager data-parallel step."""
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_graph_worker, args=(1, port, out), nprocs=1, join=True)
        r = dict(out)[0]
    assert r["cap_g"] and not r["cap_e"]
    np.testing.assert_allclose(np.array(r["graph"], dtype=np.float64), np.array(r["eager"], dtype=np.float64), rtol=2e-4)
    assert r["pdiff"] < 1e-5


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] (c4) at a batch the CPU oracle finishes in seconds: the c2 network (64x64x3, z=128, channels
# (64,128,256,512)), global batch 8 sharded over 2 processes with Sync-BN, against the oracle's single-process
# full-batch step.
C2 = dict(cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)
C2_HP = dict(beta_kl=0.5, beta_rec=0.75, beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4, n=10000)


def _c2_inputs(B=8):
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    return x, [torch.randn(B, C2["zdim"], generator=g) for _ in range(6)]


_C2_ORACLES = {}


def _c2_oracles():
    """fp32 and fp64 oracle traces of the full-batch step, computed once per session."""
    if _C2_ORACLES:
        return _C2_ORACLES
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import contextlib
    import io
    import models
    from oracle.network import Net
    from oracle.steps import Trainer
    from step_trace import traced_oracle_step
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        sd = {k: v.clone() for k, v in models.SoftIntroVAE(arch="conv", **C2).state_dict().items()}
    x, draws = _c2_inputs()
    oracles = {}
    for name, dt in (("o32", torch.float32), ("o64", torch.float64)):
        st = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        tr = Trainer("intro_tc", Net("conv", state=st, **C2), dataset_size=C2_HP["n"], beta_kl=C2_HP["beta_kl"],
                     beta_rec=C2_HP["beta_rec"], beta_neg=C2_HP["beta_neg"], gamma_r=C2_HP["gamma_r"], clip=C2_HP["clip"],
                     lr=C2_HP["lr"])
        oracles[name] = traced_oracle_step(tr, x.to(dt), [t.to(dt) for t in draws])
    _C2_ORACLES.update(oracles)
    return _C2_ORACLES


def _c2_worker(rank, world, port, out, math):
    _setup(rank, world, port, "gloo")
    import contextlib
    import io
    import models
    from hipvae import ddp
    from solvers.intro_tc import IntroTCSovler
    from step_trace import traced_hip_step
    ddp.init(sync_bn=True)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.SoftIntroVAE(arch="conv", **C2)
    model = model.to(dev).train()

    class DS:
        def __len__(self):
            return C2_HP["n"]

    x, draws = _c2_inputs()
    Bl = x.shape[0] // world
    sl = slice(rank * Bl, (rank + 1) * Bl)
    solver = IntroTCSovler(DS(), model, Bl, torch.optim.Adam(model.encoder.parameters(), lr=C2_HP["lr"]),
                           torch.optim.Adam(model.decoder.parameters(), lr=C2_HP["lr"]), "mse", C2_HP["beta_kl"],
                           C2_HP["beta_rec"], C2_HP["beta_neg"], C2_HP["gamma_r"], dev, False, None, clip=C2_HP["clip"])
    solver.conv_math = math
    tr = traced_hip_step(solver, model, x[sl], [t[sl].clone() for t in draws])
    out[rank] = dict(dict=tr["dict"], decoded=[t.numpy() for t in tr["decoded"]],
                     encoded=[[t.numpy() for t in pair] for pair in tr["encoded"]],
                     kl=[t.numpy() for t in tr["kl"]], rec=[t.numpy() for t in tr["rec"]],
                     grads=[{k: v.numpy() for k, v in g.items()} for g in tr["grads"]])
    ddp.shutdown()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["fp32", "f16x3", "bf16x3"])
def test_ddp_c2_shape_step_vs_oracle(math):
    """Two processes x 4 images, Sync-BN, all-gathered means for the TC estimator, averaged gradients == the oracle's
    single-process step on the 8 images: returned scalars, every image / encoder output of the local rows, the
    per-sample hook outputs of the local rows, and the averaged gradients of both phases."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.network import Net
    from oracle.steps import Trainer
    from step_trace import STEP_TOL, rel_err, traced_oracle_step
    import contextlib
    import io
    import models
    oracles = _c2_oracles()
    ref = oracles["o32"]
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_c2_worker, args=(world, port, out, math), nprocs=world, join=True)
        res = dict(out)
    assert res[0]["dict"] == res[1]["dict"]
    for k in ("loss_enc", "loss_dec", "loss_kl", "loss_rec", "L2"):
        t = 1e-3 if (k == "L2" and math == "bf16x3") else 1e-4
        assert abs(res[0]["dict"][k] - ref["dict"][k]) <= t * abs(ref["dict"][k]), (k, res[0]["dict"][k], ref["dict"][k])
    from step_trace import compare_traces
    for r in (0, 1):
        compare_traces(res[r], ref, oracles["o64"], STEP_TOL[math], f"ddp c2 {math} rank {r}", rows=slice(r * 4, (r + 1) * 4))
