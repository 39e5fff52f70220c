#!/usr/bin/env python3
"""Headline benchmark: images/sec of one IntroTCSovler.train_step (Soft-Intro beta-TC-VAE step:
5 encoder + 8 decoder forwards, 2 backwards, 2 Adam updates) on synthetic batches resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5] [--math bf16x3|bf16x6|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs; conv architecture, Adam lr 2e-4, clip 100, N = 10000):
    c2 (default)  64x64x3,  z=128, channels (64,128,256,512),         64 images per GPU   -- the metric's configuration;
                  with --gpus 8 this is BASELINE configs[3] (c4: global batch 512)
    c3            128x128x3, z=256, channels (64,128,256,512,512),     128 images per GPU
    c5            256x256x3, z=512, channels (64,128,256,512,512,512), 32 images per GPU  (--gpus 8: global batch 256)
fp32 tensors throughout; the conv products run in the mode --math selects (default f16x3 = what the reference's
use_amp=True config maps to: fp32-class accuracy on the fp16 matrix cores).  At N=1 on c2 the line also carries
"modes": the same measurement in exact fp32 (the reference's own precision), bf16x3 and bf16x6, made in the same process.

One process per GPU; N>1 shards the batch (weak scaling) with a flat RCCL all-reduce of the trained half's gradients
per phase and an all-gather of mu for the full-batch TC estimator; BatchNorm statistics are per rank (throughput
mode) unless --sync-bn (parity mode).  Rank 0 prints ONE JSON line.  The timed region is bracketed by barrier +
synchronize on both sides and the maximum over ranks is reported.  N=1 times K hipGraph replays of the whole step.
N>1 times K eager steps and then K replays of the captured data-parallel step (RCCL collectives inside the graph;
3.5 % faster on a one-rank RCCL group: 16.8 vs 17.4 ms) and reports the faster.  Each rank's bench.py is then a
SUPERVISOR that never touches the GPU: it runs the eager measurement in one child process (`--leg eager`, a complete
measurement on its own; its failure is this process's failure, same exit status) and the captured leg in a second,
fresh child (`--leg graph`, own rendezvous port, a deadline).  If that optional leg fails, hangs or aborts, its child
dies with its own non-zero status and the supervisor prints the eager line with "graph_leg": {"status": "abandoned",
"reason": ...}; nothing in a GPU-touched process turns a fault into status 0.  ITCV_DDP_GRAPH=0 skips the attempt.

Extra objects on the line:
  roofline      dominant kernel (an implicit-GEMM conv on the matrix cores): algorithmic FLOP of its launches / their
                HIP-event durations, measured live on the launch stream; peak = dense MFMA peak of the arithmetic
                (2500/3 TFLOP/s for f16x3 and bf16x3, 2500/6 for bf16x6, 157.3 for fp32; MI355X_MICROARCH.md); traffic = HBM
                bytes per launch from the committed rocprofv3 --pmc passes, only while the kernel sources are unchanged.
  cpu_baseline  the CPU oracle (oracle/, a PyTorch-CPU port pinned to the reference by golden vectors) timed on the
                host cores: 1 warm-up + 3 timed steps (BASELINE.md protocol) on a bounded batch of the same workload.
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "intro-tc-vae_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# 13*Fe + 19*Fd GFLOP per image per intro-tc step (2*MAC, conv + linear)
CONFIGS = {
    "c2": dict(cfg=dict(cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64), batch=64, gflop=48.237, cpu_batch=64,
               metric="images/sec (64x64x3, z=128, bs=64) intro-TC step"),
    "c3": dict(cfg=dict(cdim=3, zdim=256, channels=(64, 128, 256, 512, 512), image_size=128), batch=128, gflop=197.589,
               cpu_batch=16, metric="images/sec (128x128x3, z=256, bs=128) intro-TC step"),
    "c5": dict(cfg=dict(cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256), batch=32, gflop=794.812,
               cpu_batch=4, metric="images/sec (256x256x3, z=512, bs=32/GPU) intro-TC step"),
}
HP = dict(beta_kl=0.5, beta_rec=0.75, beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4, dataset=10000)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense; a split product costs 3 (bf16x3) or 6 (bf16x6) bf16 MFMA products
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
PEAK_HBM_GBS = 8000.0                # HBM3E spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s reachable with a float4 copy)
DTYPE = {"fp32": "f32", "bf16x3": "f32 in/out/accumulate, conv products on split-bf16 MFMA (bf16x3)",
         "bf16x6": "f32 in/out/accumulate, conv products on split-bf16 MFMA (bf16x6, fp32-class)",
         "f16x3": "f32 in/out/accumulate, conv products on split-fp16 MFMA (two scaled fp16 planes, 3 products, fp32-class)"}


class _DS:
    def __len__(self):
        return HP["dataset"]


def csrc_digest():
    """sha256 over the kernel sources: the PMC traffic figures are valid for exactly these."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "intro-tc-vae_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the
    GPU box hands a one-GPU job a 16-core share of a much larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("ITCV_CPU_THREADS", n))))


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def quiet_model(cfg):
    import models
    with contextlib.redirect_stdout(io.StringIO()):       # the reference prints the conv shape at construction
        return models.SoftIntroVAE(arch="conv", **cfg)


def cpu_baseline(wl):
    """Times the CPU oracle's intro-TC step on this host: 1 warm-up + 3 timed steps at ``cpu_batch``
    images (the full per-GPU batch for c2; a bounded sample of it for the larger workloads)."""
    from oracle.network import Net
    from oracle.steps import Trainer
    cfg, B = wl["cfg"], wl["cpu_batch"]
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
    torch.manual_seed(0)
    state = {k: v.clone() for k, v in quiet_model(cfg).state_dict().items()}
    tr = Trainer("intro_tc", Net("conv", state=state, **cfg), dataset_size=HP["dataset"], beta_kl=HP["beta_kl"],
                 beta_rec=HP["beta_rec"], beta_neg=HP["beta_neg"], gamma_r=HP["gamma_r"], clip=HP["clip"], lr=HP["lr"])
    g = torch.Generator().manual_seed(1234)

    def one():
        x = torch.rand(B, 3, cfg["image_size"], cfg["image_size"], generator=g)
        draws = [torch.randn(B, cfg["zdim"], generator=g) for _ in range(6)]
        t0 = time.perf_counter()
        tr.step(x, draws)
        return time.perf_counter() - t0

    log(f"cpu warm-up step B={B}: {one():.1f} s")
    ts = [one() for _ in range(3)]
    log(f"cpu timed steps B={B}: " + ", ".join(f"{t:.1f} s" for t in ts))
    return {"value": round(3 * B / sum(ts), 3), "unit": "images/s", "cores": cores, "kind": "port",
            "step_seconds": [round(t, 2) for t in ts],
            "sample": f"3 timed intro-tc steps at B={B} after 1 warm-up step (BASELINE.md protocol); "
                      "oracle/ (PyTorch-CPU fp32 port of the reference step, pinned by golden vectors)"}


def make_solver(wl, math, dev):
    from solvers.intro_tc import IntroTCSovler
    torch.manual_seed(0)
    model = quiet_model(wl["cfg"]).to(dev).train()
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=HP["lr"])
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=HP["lr"])
    solver = IntroTCSovler(_DS(), model, wl["batch"], opt_e, opt_d, "mse", HP["beta_kl"], HP["beta_rec"],
                           HP["beta_neg"], HP["gamma_r"], dev, math != "fp32", None, clip=HP["clip"])
    solver.conv_math = math
    return solver


def _pmc(cfg_name, math):
    """PMC record of this (workload, arithmetic) from profiles/r03_pmc_traffic.json, or (None, reason)."""
    try:
        pmc = json.load(open(PMC_FILE))
    except Exception:  # noqa: BLE001
        return None, "profiles/r03_pmc_traffic.json not found"
    rec = pmc.get("configs", {}).get(cfg_name)
    if not rec or rec.get("math") != math:
        return None, f"no PMC record for workload {cfg_name} / {math}"
    if pmc.get("csrc_sha256") != csrc_digest():
        return None, ("stale: the kernel sources changed since profiles/r03_pmc_traffic.json was collected (csrc_sha256 "
                      "differs) -- re-run tools/evidence.sh")
    return rec, ""


def roofline_of(records, math, steps, eager_elapsed, where, cfg_name="c2"):
    """Dominant-kernel rooflines from the live HIP-event records of the eager leg: the matrix-core one over the conv /
    weight-gradient launches (work = algorithmic FLOP) and -- "hbm" -- the HBM one over the BatchNorm apply passes (work =
    algorithmic bytes)."""
    hbm_names = ("bn_act_fwd_planes_kernel", "bn_bwd_apply_planes")
    buckets, hbuckets = {}, {}
    for label, work, secs in records:
        b = (hbuckets if label.startswith(hbm_names) else buckets).setdefault(label, [0, 0.0, 0.0])
        b[0] += 1
        b[1] += work
        b[2] += secs
    conv_time = sum(b[2] for b in buckets.values())
    conv_flop = sum(b[1] for b in buckets.values())
    dom_label, dom = max(buckets.items(), key=lambda kv: kv[1][2])
    achieved = dom[1] / dom[2] * 1e-12
    if "bf16s" in dom_label or "bf16p" in dom_label or "planes" in dom_label or "mfma" in dom_label:
        products = 6 if "NS=3" in dom_label else 3
        kind = "fp16" if "NS=4" in dom_label else "bf16"
        peak, peak_note = PEAK_BF16_MFMA_TFLOPS / products, f"2500 TFLOP/s dense {kind} MFMA / {products} {kind} products per fp32 product"
    else:
        peak, peak_note = PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA"
    pmc, why = _pmc(cfg_name, math)
    traffic, traffic_note = None, why or "no PMC record for this kernel"
    rec = pmc["kernels"].get(dom_label) if pmc else None
    if rec:
        traffic = rec["hbm_bytes_per_launch_fetch_x2"]
        traffic_note = (f"{os.path.relpath(PMC_FILE, ROOT)} [{cfg_name}]: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, rocprofv3 --pmc, "
                        f"separate passes, same kernel sources (csrc_sha256); raw (uncorrected) = {rec['hbm_bytes_per_launch_raw']}; "
                        "algorithmic operand + result bytes per launch: B*H*W*(4*Ci + 4*Co) + packed weights (DESIGN.md section 6)")
    out = {
        "bound": "mfma", "kernel": dom_label, "achieved": round(achieved, 2), "peak": round(peak, 1),
        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
        "peak_note": peak_note, "launches_per_step": dom[0] / steps, "avg_launch_us": round(dom[2] / dom[0] * 1e6, 2),
        "algorithmic_gflop_per_launch": round(dom[1] / dom[0] * 1e-9, 3),
        "all_conv_kernels": {"achieved": round(conv_flop / conv_time * 1e-12, 2),
                             "share_of_eager_step_time": round(conv_time / eager_elapsed, 3)},
        "buckets": [{"kernel": k, "ms_per_step": round(b[2] / steps * 1e3, 3), "launches_per_step": b[0] / steps,
                     "tflops": round(b[1] / b[2] * 1e-12, 1)}
                    for k, b in sorted(buckets.items(), key=lambda kv: -kv[1][2])[:14]],
        "measured": "HIP event pairs stamped at kernel start/end (hipExtLaunchKernelGGL inside libitcv_hip.so, launch "
                    f"stream) for every main conv kernel during {steps} eager steps of this workload, same process; they "
                    "include the end-of-kernel L2 write-back of the result, which rocprofv3's dispatch timestamps do not "
                    "(5-20 % shorter there)" + where,
    }
    if hbuckets:
        # the HBM-bound block of the step (BatchNorm: ~27 % of kernel time): its largest bucket against the HBM peak
        hl, hb = max(hbuckets.items(), key=lambda kv: kv[1][2])
        gbs = hb[1] / hb[2] * 1e-9
        base = hl.split("<")[0]
        htraffic, hnote = None, why or "no PMC record for this kernel"
        hrec = pmc["kernels"].get(base) if pmc else None
        # the rocprofv3 name carries no shape: the record's LARGEST launch of this kernel is this bucket's launch only if
        # the bucket is the kernel's largest one (by algorithmic bytes per launch)
        biggest = max((k for k in hbuckets if k.startswith(base + "<")), key=lambda k: hbuckets[k][1] / hbuckets[k][0])
        if hrec and biggest == hl:
            htraffic = hrec["max_launch"]["hbm_bytes_fetch_x2"]
            hnote = (f"{os.path.relpath(PMC_FILE, ROOT)} [{cfg_name}]: the kernel's largest launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "(the two passes paired by launch ordinal)")
        elif hrec:
            hnote = "the PMC record's largest launch of this kernel is another shape"
        out["hbm"] = {"bound": "hbm", "kernel": hl, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": htraffic, "traffic_note": hnote,
                      "avg_launch_us": round(hb[2] / hb[0] * 1e6, 2), "launches_per_step": hb[0] / steps,
                      "algorithmic_mb_per_launch": round(hb[1] / hb[0] * 1e-6, 2),
                      "share_of_eager_step_time": round(sum(b[2] for b in hbuckets.values()) / eager_elapsed, 3),
                      "note": "BatchNorm apply pass (reads the conv output [and its gradient] once more, writes the planes): "
                              "achieved = algorithmic bytes / live HIP-event time; right behind its producer much of the "
                              "input is served by the 256 MiB Infinity Cache, so `achieved` can exceed what HBM alone delivers"}
    return out


def measure(wl, math, dev, args, rank, world, sync, use_graph, ddp_on, with_h2d=False):
    """Warm-up, roofline leg (K eager steps with per-launch events), timed leg.  Returns a dict of raw results."""
    from hipvae import functional as HF
    solver = make_solver(wl, math, dev)
    B, S = wl["batch"], wl["cfg"]["image_size"]
    g = torch.Generator().manual_seed(1000 + rank)
    batches = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(4)]   # resident in HBM
    last = None
    for i in range(args.warmup):
        last = solver.train_step(batches[i % len(batches)], i)
        if rank == 0:
            log(f"[{math}] warm-up step {i}: {last}")
    sync()
    # ---- roofline leg: K eager steps with a HIP-event pair around every implicit-GEMM launch ----
    HF.LaunchProfile.begin()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = solver.train_step(batches[i % len(batches)], args.warmup + i)
    sync()
    eager_elapsed = time.perf_counter() - t0
    records = HF.LaunchProfile.end()
    elapsed = eager_elapsed
    if rank == 0:
        log(f"[{math}] eager: {args.steps} steps in {eager_elapsed:.3f} s (events on)")
    if use_graph:
        # ---- timed region proper: the same K steps as hipGraph replays (one submission per step) ----
        solver.enable_graph()
        for i in range(5):                      # 3 eager warm-ups, capture + first replay, one more replay
            last = solver.train_step(batches[i % len(batches)], 0)
        assert solver._graph is not None
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = solver.train_step(batches[i % len(batches)], args.warmup + args.steps + i)
        sync()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            log(f"[{math}] graph: {args.steps} steps in {elapsed:.3f} s")
    elif ddp_on:
        # ---- timed region proper at N>1: K eager steps without the per-launch event pairs ----
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = solver.train_step(batches[i % len(batches)], args.warmup + args.steps + i)
        sync()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            log(f"[{math}] eager, events off: {args.steps} steps in {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    h2d = None
    if world == 1 and with_h2d:
        # ---- the same K steps with every batch starting in pageable HOST memory (what a DataLoader hands over):
        # hipvae.loader.PrefetchLoader stages it in pinned memory and copies it on a side stream while the previous
        # step runs.  Reported beside `value`, never as `value` (which has its inputs resident in HBM).
        from hipvae.loader import PrefetchLoader
        host = [b.cpu() for b in batches]

        class _Feed:
            def __len__(self):
                return args.steps

            def __iter__(self):
                return (host[i % len(host)] for i in range(args.steps))

        loader = PrefetchLoader(_Feed(), None, device=dev)
        for _ in range(2):                       # first pass allocates the pinned / device rings
            sync()
            t0 = time.perf_counter()
            for i, xb in enumerate(loader):
                last = solver.train_step(xb, args.warmup + 2 * args.steps + i)
            sync()
            t_h2d = time.perf_counter() - t0
        h2d = {"value": round(B * args.steps / t_h2d, 2), "unit": "images/s", "ms_per_step": round(t_h2d / args.steps * 1e3, 3),
               "bytes_per_step": int(batches[0].numel() * 4),
               "note": "batches start in pageable host memory; hipvae.loader.PrefetchLoader (pinned ring, H2D on a side "
                       "stream under the previous step); same execution mode as `value`"}
        if rank == 0:
            log(f"[{math}] host-fed: {args.steps} steps in {t_h2d:.3f} s")
    where = (", immediately before the timed hipGraph-replay steps" if use_graph else
             ", immediately before the timed eager steps" if ddp_on else " (the timed region)")
    return dict(solver=solver, batches=batches, elapsed=elapsed, eager_elapsed=eager_elapsed, last=last, h2d=h2d,
                roofline=roofline_of(records, math, args.steps, eager_elapsed, where, args.config))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="workload: c2 = the metric's configuration (c4 when --gpus 8); c3 / c5 = BASELINE configs[2] / [4]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true",
                    help="N=1, c2: skip the additional fp32 / bf16x6 measurements (the 'modes' object)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1: all-reduce BatchNorm moments (full-batch parity mode); default is per-rank statistics")
    ap.add_argument("--math", choices=["f16x3", "bf16x3", "bf16x6", "fp32"], default="f16x3",
                    help="conv GEMM arithmetic: f16x3 = use_amp=True (the reference's config default): two scaled fp16 "
                         "planes, 3 MFMA products, fp32 accumulate, fp32-class accuracy; bf16x3 = two bf16 planes "
                         "(2^-16 per product); bf16x6 = three bf16 planes, 6 products; fp32 = exact fp32 MFMA")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replays (N=1)")
    ap.add_argument("--leg", choices=["eager", "graph"], default=None,
                    help="(internal, N>1) run ONE leg of the data-parallel measurement in this process; without it an N>1 "
                         "invocation supervises the two legs as child processes")
    args = ap.parse_args()
    wl = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    # ITCV_BENCH_BACKEND=gloo (test rigs with fewer GPUs than ranks: ranks share devices, collectives staged on
    # the host) -- the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("ITCV_BENCH_BACKEND", "nccl")
    force_ddp = world == 1 and os.environ.get("ITCV_BENCH_FORCE_DDP", "0") == "1"
    if ((world > 1 or force_ddp) and args.leg is None and not args.no_graph and backend == "nccl"
            and os.environ.get("ITCV_DDP_GRAPH", "try") == "try"):
        return _supervise_legs(rank)             # this process never initialises the GPU
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ITCV_BENCH_FORCE_DDP=1 (one-GPU rigs): a group of ONE rank with the data-parallel code paths on, so the step's
    # collectives run through RCCL on the device -- the launch-side cost of the N>1 path without the wire time
    if force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        with _stdout_to_stderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            dist.barrier()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():              # RCCL prints its version banner on stdout when the communicator is made
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()

    from hipvae import ddp

    if world > 1 or force_ddp:
        ddp.init(sync_bn=args.sync_bn, force=force_ddp)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.leg == "graph":
        return _graph_leg(wl, args, dev, rank, world, sync)
    # N>1: eager unless ITCV_DDP_GRAPH=1 (RCCL collectives captured in the step graph; hipvae.ddp.graph_capturable)
    ddp_on = ddp.get() is not None
    use_graph = not args.no_graph and (not ddp_on or ddp.graph_capturable())
    B = wl["batch"]
    m = measure(wl, args.math, dev, args, rank, world, sync, use_graph, ddp_on, with_h2d=not ddp_on)
    solver, batches, elapsed, eager_elapsed, last = m["solver"], m["batches"], m["elapsed"], m["eager_elapsed"], m["last"]

    images = B * world * args.steps
    value = images / elapsed
    cfg = wl["cfg"]
    out = {
        "metric": wl["metric"], "value": round(value, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE[args.math], "data": "synthetic",
        "config": {"workload": f"{args.config}: IntroTCSovler.train_step, conv arch, {cfg['image_size']}x{cfg['image_size']}x3, "
                               f"z_dim={cfg['zdim']}, channels {tuple(cfg['channels'])}, batch {B}/GPU, Adam lr 2e-4, clip 100, N=10000",
                   "global_batch": B * world,
                   "parallelism": f"dp{world}" if world > 1 else ("single, dp code path forced (one-rank RCCL group)" if force_ddp else "single"),
                   "sync_bn": bool(ddp_on and args.sync_bn)},
        "execution": "hipGraph replay (whole step = one graph)" if use_graph else "eager launches",
        "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 3),
        "step_tflop": round(wl["gflop"] * B * world * 1e-3, 3),
        "whole_step_tflops": round(value * wl["gflop"] * 1e-3 / world, 2),
        "last_step": last,
        "roofline": m["roofline"],
    }
    if m["h2d"] is not None:
        out["host_fed"] = m["h2d"]
    if ddp_on:
        out["graph_leg"] = {"status": "main" if use_graph else ("pending" if args.leg == "eager" else "skipped"),
                            "reason": "ITCV_DDP_GRAPH=" + os.environ.get("ITCV_DDP_GRAPH", "") + (" / --no-graph" if args.no_graph else "")
                            + (" / backend " + backend if backend != "nccl" else "")}
    # ---- N=1 on the metric's workload: the same measurement in the reference's own precision (and bf16x6) ----------
    if world == 1 and not ddp_on and args.config == "c2" and not args.no_modes:
        del solver, batches, m
        torch.cuda.empty_cache()
        out["modes"] = {}
        for math in ("fp32", "bf16x3", "bf16x6"):
            if math == args.math:
                continue
            mm = measure(wl, math, dev, args, rank, world, sync, use_graph, ddp_on)
            v = B * args.steps / mm["elapsed"]
            r = mm["roofline"]
            out["modes"][math] = {"value": round(v, 2), "unit": "images/s", "ms_per_step": round(mm["elapsed"] / args.steps * 1e3, 3),
                                  "dtype": DTYPE[math], "whole_step_tflops": round(v * wl["gflop"] * 1e-3, 2),
                                  "last_step": mm["last"],
                                  "roofline": {k: r[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                                 "launches_per_step")}}
            del mm
            torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()


@contextlib.contextmanager
def _stdout_to_stderr():
    """File-descriptor level: native libraries' stdout goes to stderr inside the block (stdout carries the JSON line)."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _graph_leg(wl, args, dev, rank, world, sync):
    """`--leg graph` (child process of the supervisor): the data-parallel step captured into one hipGraph per rank, RCCL
    collectives included; times K replays (max over ranks); rank 0 prints {"graph_elapsed_s": ...}.  Any failure is this
    process's own failure (non-zero exit status / signal); the supervisor decides what to report."""
    fault = os.environ.get("ITCV_BENCH_GRAPH_FAULT")          # test hook: "hang" | "raise" | "abort"
    if fault == "abort":
        os.abort()
    if fault == "hang":
        time.sleep(1e6)
    if fault == "raise":
        raise RuntimeError("injected fault")
    solver = make_solver(wl, args.math, dev)
    B, S = wl["batch"], wl["cfg"]["image_size"]
    g = torch.Generator().manual_seed(1000 + rank)
    batches = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(4)]
    solver.enable_graph()
    for i in range(max(args.warmup, 5)):         # 3 eager warm-ups, capture + first replay, further replays
        solver.train_step(batches[i % len(batches)], i)
    if solver._graph is None:
        raise RuntimeError("the data-parallel step was not captured")
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        solver.train_step(batches[i % len(batches)], args.warmup + i)
    sync()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        log(f"graph (data-parallel): {args.steps} steps in {float(t):.3f} s")
        print(json.dumps({"graph_elapsed_s": float(t), "steps": args.steps}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def _supervise_legs(rank):
    """N>1 (or the one-rank rig ITCV_BENCH_FORCE_DDP=1): run the two legs as child processes of this one, which never
    initialises the GPU.  Leg 1 (eager) IS the measurement: if it fails, so does this process, with the child's status.
    Leg 2 (captured) is optional: whatever happens to its process -- exception, abort() in RCCL's watchdog thread, a
    rank stuck in a collective until the deadline -- rank 0 reports the finished eager line with the reason."""
    import subprocess
    me = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    a = subprocess.run(me + ["--leg", "eager"], stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in a.stdout.splitlines() if ln.startswith("{")]
    if a.returncode != 0 or (rank == 0 and not lines):
        sys.stdout.write(a.stdout)
        log(f"eager leg failed (status {a.returncode})")
        sys.exit(a.returncode if a.returncode > 0 else 1)
    out = json.loads(lines[-1]) if rank == 0 else None
    env = dict(os.environ)
    # own rendezvous: a store of this leg's own on the next port (the launcher's agent store still holds the first
    # leg's communicator keys)
    env["MASTER_ADDR"] = env.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29541")) + 1)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    env["ITCV_DDP_GRAPH"] = "1"
    deadline = float(os.environ.get("ITCV_BENCH_GRAPH_DEADLINE", "240"))
    status, reason, g_elapsed = "abandoned", None, None
    try:
        b = subprocess.run(me + ["--leg", "graph"], stdout=subprocess.PIPE, text=True, env=env, timeout=deadline)
        glines = [ln for ln in b.stdout.splitlines() if ln.startswith("{")]
        if b.returncode == 0 and (rank != 0 or glines):
            status = "ok"
            if rank == 0:
                g_elapsed = json.loads(glines[-1])["graph_elapsed_s"]
        else:
            reason = (f"the captured leg's process ended with status {b.returncode}"
                      + (" (killed by signal %d)" % -b.returncode if b.returncode < 0 else ""))
    except subprocess.TimeoutExpired:
        reason = f"no result within {deadline:.0f} s (process killed)"
    if rank != 0:
        return
    if status == "ok":
        steps, images = out["steps"], out["config"]["global_batch"] * out["steps"]
        out["graph_leg"] = {"status": "ok", "ms_per_step": round(g_elapsed / steps * 1e3, 3)}
        out["eager_events_off_ms_per_step"] = out["ms_per_step"]
        out["graph_ms_per_step"] = round(g_elapsed / steps * 1e3, 3)
        if out["graph_ms_per_step"] < out["ms_per_step"]:
            eager_value, value = out["value"], images / g_elapsed
            out.update(value=round(value, 2), ms_per_step=out["graph_ms_per_step"],
                       execution="hipGraph replay (whole data-parallel step, RCCL collectives included, = one graph per rank)",
                       whole_step_tflops=round(out["whole_step_tflops"] * value / eager_value, 2))
    else:
        log(f"captured data-parallel leg abandoned ({reason}); reporting the eager measurement")
        out["graph_leg"] = {"status": "abandoned", "reason": reason, "deadline_s": deadline}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
