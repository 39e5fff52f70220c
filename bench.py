#!/usr/bin/env python3
"""Headline benchmark: images/sec of one IntroTCSovler.train_step (Soft-Intro beta-TC-VAE step:
5 encoder + 8 decoder forwards, 2 backwards, 2 Adam updates) on synthetic 64x64x3 batches,
z_dim=128, batch 64 per GPU, conv architecture -- BASELINE.json configs[1] (c2).  fp32 tensors; the conv
products run in the mode --math selects (default bf16x3 = what the reference's use_amp=True config maps to).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; N>1 shards the batch (weak scaling, 64 images per GPU) with a flat RCCL
all-reduce of the trained half's gradients per phase and an all-gather of mu for the full-batch TC
estimator; BatchNorm statistics are per rank (throughput mode) unless --sync-bn (parity mode).  Rank 0 prints ONE JSON line.  Inputs are resident in HBM before
the timed region; the timed region is bracketed by barrier + synchronize on both sides and the
maximum over ranks is reported.  N=1 times K hipGraph replays of the whole step.  N>1 times K eager steps, then K
replays of the captured data-parallel step (RCCL collectives inside the graph) under a watchdog that falls back to
the finished eager measurement, and reports the faster execution (see _guarded_graph_leg; ITCV_DDP_GRAPH=0 skips it).

Extra objects on the line:
  roofline      dominant kernel (an implicit-GEMM conv on the matrix cores): algorithmic FLOP of its
                launches / their HIP-event durations, measured live on the launch stream; peak = dense
                MFMA peak of the arithmetic (2500/3 TFLOP/s for bf16x3, 157.3 for fp32; MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (oracle/, a PyTorch-CPU port pinned to the reference by golden
                vectors) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import contextlib
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

CFG = dict(cdim=3, zdim=128, channels=(64, 128, 256, 512), image_size=64)
B_PER_GPU = 64
HP = dict(beta_kl=0.5, beta_rec=0.75, beta_neg=512.0, gamma_r=1e-8, clip=100.0, lr=2e-4, dataset=10000)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense; a split product costs 3 (bf16x3) or 6 (bf16x6) bf16 MFMA products
# 13*Fe + 19*Fd = 48.237 GFLOP per image per intro-tc step (2*MAC, conv+linear)
STEP_GFLOP_PER_IMAGE = 48.237


class _DS:
    def __len__(self):
        return HP["dataset"]


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


def cpu_baseline():
    """Times the CPU oracle's intro-TC step on this host: a B=16 warm-up step, then ONE timed
    step at the full B=64 (about 10-20 s of CPU work on 8-16 cores)."""
    import models
    from oracle.network import Net
    from oracle.steps import Trainer
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
    torch.manual_seed(0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.SoftIntroVAE(arch="conv", **CFG)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    tr = Trainer("intro_tc", Net("conv", state=state, **CFG), dataset_size=HP["dataset"], beta_kl=HP["beta_kl"],
                 beta_rec=HP["beta_rec"], beta_neg=HP["beta_neg"], gamma_r=HP["gamma_r"], clip=HP["clip"], lr=HP["lr"])
    g = torch.Generator().manual_seed(1234)

    def one(batch):
        x = torch.rand(batch, 3, 64, 64, generator=g)
        draws = [torch.randn(batch, CFG["zdim"], generator=g) for _ in range(6)]
        t0 = time.perf_counter()
        tr.step(x, draws)
        return time.perf_counter() - t0

    log(f"cpu warm-up step B=16: {one(16):.1f} s")
    t = one(B_PER_GPU)
    log(f"cpu timed step B={B_PER_GPU}: {t:.1f} s")
    return {"value": round(B_PER_GPU / t, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 intro-tc step at B={B_PER_GPU} ({t:.1f} s) after a B=16 warm-up step; "
                      "oracle/ (PyTorch-CPU fp32 port of the reference step, pinned by golden vectors)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1: all-reduce BatchNorm moments (full-batch parity mode); default is per-rank statistics")
    ap.add_argument("--math", choices=["bf16x3", "bf16x6", "fp32"], default="bf16x3",
                    help="conv GEMM arithmetic: bf16x3 = use_amp=True (the reference's config default), split-bf16 "
                         "MFMA with fp32 accumulate; bf16x6 = fp32-class 3-way split; fp32 = exact fp32 MFMA")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replays (N=1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    # ITCV_BENCH_BACKEND=gloo (test rigs with fewer GPUs than ranks: ranks share devices, collectives staged on
    # the host) -- the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("ITCV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ITCV_BENCH_FORCE_DDP=1 (one-GPU rigs): a group of ONE rank with the data-parallel code paths on, so the step's
    # collectives run through RCCL on the device -- the launch-side cost of the N>1 path without the wire time
    force_ddp = world == 1 and os.environ.get("ITCV_BENCH_FORCE_DDP", "0") == "1"
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

    import contextlib
    import io
    import models
    from hipvae import ddp
    from hipvae import functional as HF
    from solvers.intro_tc import IntroTCSovler

    if world > 1 or force_ddp:
        ddp.init(sync_bn=args.sync_bn, force=force_ddp)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.SoftIntroVAE(arch="conv", **CFG)
    model = model.to(dev).train()
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=HP["lr"])
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=HP["lr"])
    solver = IntroTCSovler(_DS(), model, B_PER_GPU, opt_e, opt_d, "mse", HP["beta_kl"], HP["beta_rec"],
                           HP["beta_neg"], HP["gamma_r"], dev, args.math != "fp32", None, clip=HP["clip"])
    solver.conv_math = args.math
    g = torch.Generator().manual_seed(1000 + rank)
    batches = [torch.rand(B_PER_GPU, 3, 64, 64, generator=g).to(dev) for _ in range(4)]   # resident in HBM

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N>1: eager unless ITCV_DDP_GRAPH=1 (RCCL collectives captured in the step graph; hipvae.ddp.graph_capturable)
    use_graph = not args.no_graph and (ddp.get() is None or ddp.graph_capturable())
    last = None
    for i in range(args.warmup):
        last = solver.train_step(batches[i % len(batches)], i)
        if rank == 0:
            log(f"warm-up step {i}: {last}")
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
        log(f"eager: {args.steps} steps in {eager_elapsed:.3f} s (events on)")
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
            log(f"graph: {args.steps} steps in {elapsed:.3f} s")
    elif ddp.get() is not None:
        # ---- timed region proper at N>1: K eager steps without the per-launch event pairs ----
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = solver.train_step(batches[i % len(batches)], args.warmup + args.steps + i)
        sync()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            log(f"eager, events off: {args.steps} steps in {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # ---- roofline of the dominant kernel from the live HIP-event records -------------------
    buckets = {}
    for label, flop, secs in records:
        b = buckets.setdefault(label, [0, 0.0, 0.0])
        b[0] += 1
        b[1] += flop
        b[2] += secs
    conv_time = sum(b[2] for b in buckets.values())
    conv_flop = sum(b[1] for b in buckets.values())
    dom_label, dom = max(buckets.items(), key=lambda kv: kv[1][2])
    achieved = dom[1] / dom[2] * 1e-12
    if "bf16s" in dom_label or "bf16p" in dom_label:
        products = 3 if "NS=2" in dom_label else 6
        peak, peak_note = PEAK_BF16_MFMA_TFLOPS / products, f"2500 TFLOP/s dense bf16 MFMA / {products} bf16 products per fp32 product"
    else:
        peak, peak_note = PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA"
    traffic, traffic_note = None, None
    try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes over this same command (tools/pmc_summary.py)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        rec = pmc["kernels"].get(dom_label) if args.math == pmc.get("math") else None
        if rec:
            traffic = rec["hbm_bytes_per_launch_fetch_x2"]
            traffic_note = ("profiles/r01_pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, rocprofv3 --pmc, "
                            f"separate passes; raw (uncorrected) = {rec['hbm_bytes_per_launch_raw']}; algorithmic operand + "
                            "result bytes per launch: B*H*W*(4*Ci + 4*Co) + packed weights (DESIGN.md section 6)")
    except Exception:  # noqa: BLE001
        pass
    roofline = {
        "bound": "mfma", "kernel": dom_label, "achieved": round(achieved, 2), "peak": round(peak, 1),
        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note, "peak_note": peak_note,
        "launches_per_step": dom[0] / args.steps, "avg_launch_us": round(dom[2] / dom[0] * 1e6, 2),
        "algorithmic_gflop_per_launch": round(dom[1] / dom[0] * 1e-9, 3),
        "all_conv_kernels": {"achieved": round(conv_flop / conv_time * 1e-12, 2),
                             "share_of_eager_step_time": round(conv_time / eager_elapsed, 3)},
        "measured": f"HIP event pairs stamped at kernel start/end (hipExtLaunchKernelGGL inside libitcv_hip.so, launch stream) for every main conv kernel during {args.steps} eager steps of this workload, same process; they include the end-of-kernel L2 write-back of the result (about output bytes / 5 TB/s), which rocprofv3's dispatch timestamps in profiles/r01_final_kernel_stats.csv do not (5-20 % shorter there)"
                    + (", immediately before the timed hipGraph-replay steps" if use_graph else
                       ", immediately before the timed eager steps" if ddp.get() is not None else " (the timed region)"),
    }

    images = B_PER_GPU * world * args.steps
    value = images / elapsed
    out = {
        "metric": "images/sec (64x64x3, z=128, bs=64) intro-TC step", "value": round(value, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16x3": "f32 in/out/accumulate, conv products on split-bf16 MFMA (bf16x3)",
                  "bf16x6": "f32 in/out/accumulate, conv products on split-bf16 MFMA (bf16x6, fp32-class)"}[args.math],
        "data": "synthetic",
        "config": {"workload": "c2: IntroTCSovler.train_step, conv arch, 64x64x3, z_dim=128, channels (64,128,256,512), "
                               f"batch {B_PER_GPU}/GPU, Adam lr 2e-4, clip 100, N=10000",
                   "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}" if world > 1 else ("single, dp code path forced (one-rank RCCL group)" if force_ddp else "single"),
                   "sync_bn": bool(ddp.get() is not None and args.sync_bn)},
        "execution": "hipGraph replay (whole step = one graph)" if use_graph else "eager launches",
        "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 3),
        "step_tflop": round(STEP_GFLOP_PER_IMAGE * B_PER_GPU * world * 1e-3, 3),
        "whole_step_tflops": round(value * STEP_GFLOP_PER_IMAGE * 1e-3 / world, 2),
        "last_step": last,
        "roofline": roofline,
    }
    # ---- N>1 over RCCL: the same K steps once more with the data-parallel step captured into the step hipGraph ----
    # The eager measurement above is complete and stays the fallback: a watchdog prints it and ends the process
    # if the captured leg raises or makes no progress (ITCV_DDP_GRAPH=0 skips the attempt, =1 uses the graph for
    # the main timed region instead).  The faster of the two executions is the reported one, named in "execution".
    if (ddp.get() is not None and not use_graph and not args.no_graph and backend == "nccl"
            and os.environ.get("ITCV_DDP_GRAPH", "auto") == "auto"):
        g_elapsed = _guarded_graph_leg(solver, batches, args, rank, sync, dev, json.dumps(out))
        out["eager_events_off_ms_per_step"] = out["ms_per_step"]
        out["graph_ms_per_step"] = round(g_elapsed / args.steps * 1e3, 3)
        if g_elapsed < elapsed:
            value = images / g_elapsed
            out.update(value=round(value, 2), ms_per_step=out["graph_ms_per_step"],
                       execution="hipGraph replay (whole data-parallel step, RCCL collectives included, = one graph per rank)",
                       whole_step_tflops=round(value * STEP_GFLOP_PER_IMAGE * 1e-3 / world, 2))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
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


def _guarded_graph_leg(solver, batches, args, rank, sync, dev, eager_line):
    """Capture the data-parallel step and time K replays (max over ranks).  Never returns on failure: on an exception
    or when the deadline passes (a rank stuck in a collective), rank 0 prints ``eager_line`` -- the finished eager
    measurement -- and every rank that notices ends its process with status 0."""
    import threading
    deadline = float(os.environ.get("ITCV_BENCH_GRAPH_DEADLINE", "90"))
    lock, state = threading.Lock(), {"closed": False}

    def bail(reason):
        with lock:
            if state["closed"]:
                return
            state["closed"] = True
        if rank == 0:
            log(f"captured data-parallel leg abandoned ({reason}); reporting the eager measurement")
            print(eager_line, flush=True)
        sys.stderr.flush()
        os._exit(0)

    timer = threading.Timer(deadline, bail, args=(f"no result within {deadline:.0f} s",))
    timer.daemon = True
    timer.start()
    try:
        fault = os.environ.get("ITCV_BENCH_GRAPH_FAULT")          # test hook: "hang" | "raise"
        if fault == "hang":
            time.sleep(1e6)
        if fault == "raise":
            raise RuntimeError("injected fault")
        os.environ["ITCV_DDP_GRAPH"] = "1"
        solver.enable_graph()
        for i in range(5):                      # 3 eager warm-ups, capture + first replay, one more replay
            solver.train_step(batches[i % len(batches)], 0)
        if solver._graph is None:
            raise RuntimeError("the data-parallel step was not captured")
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            solver.train_step(batches[i % len(batches)], args.warmup + 2 * args.steps + i)
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g_elapsed = float(t)
    except BaseException as e:  # noqa: BLE001 -- the eager line must still come out
        bail(repr(e))
        time.sleep(1e6)                          # a concurrent bail() is ending the process
    with lock:
        closing = state["closed"]
        state["closed"] = True
    if closing:
        time.sleep(1e6)
    timer.cancel()
    if rank == 0:
        log(f"graph (data-parallel): {args.steps} steps in {g_elapsed:.3f} s")
    return g_elapsed


if __name__ == "__main__":
    main()
