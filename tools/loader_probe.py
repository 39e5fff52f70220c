#!/usr/bin/env python3
"""Where does a host-fed step spend its time?  c2 workload, hipGraph replay; batches resident / via PrefetchLoader."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from hipvae import loader as L  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.CONFIGS["c2"]
solver = bench.make_solver(wl, "bf16x3", dev)
solver.enable_graph()
xs = [torch.rand(64, 3, 64, 64) for _ in range(4)]
xd = [x.to(dev) for x in xs]
for i in range(8):
    solver.train_step(xd[i % 4], i)
torch.cuda.synchronize()
K = 20


def timed(label, it):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gaps, steps, t_prev = [], [], time.perf_counter()
    for x in it:
        t1 = time.perf_counter()
        solver.train_step(x, 0)
        t2 = time.perf_counter()
        gaps.append(t1 - t_prev), steps.append(t2 - t1)
        t_prev = t2
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{label:28s} {dt*1e3:7.2f} ms/step   fetch {sum(gaps[2:])/len(gaps[2:])*1e3:6.2f} ms  step {sum(steps[2:])/len(steps[2:])*1e3:6.2f} ms", flush=True)


class Feed:
    def __len__(self):
        return K

    def __iter__(self):
        return (xs[i % 4] for i in range(K))


timed("resident", (xd[i % 4] for i in range(K)))
timed("x.to(dev) pageable", (xs[i % 4].to(dev) for i in range(K)))
pin = [x.pin_memory() for x in xs]
timed("x.to(dev) pinned", (pin[i % 4].to(dev, non_blocking=True) for i in range(K)))
for depth in (2, 3, 4):
    ld = L.PrefetchLoader(Feed(), None, device=dev, depth=depth)
    timed(f"PrefetchLoader depth {depth} (1)", ld)
    timed(f"PrefetchLoader depth {depth} (2)", ld)

# ---- what exactly slows the step down?  variants ------------------------------------------------------------
import threading

side = torch.cuda.Stream(device=dev)
devbuf = [torch.empty_like(xd[0]) for _ in range(4)]


def v_side_stream_no_thread():          # H2D for the NEXT batch issued on a side stream right before this step
    evs = [torch.cuda.Event() for _ in range(4)]
    with torch.cuda.stream(side):
        devbuf[0].copy_(pin[0], non_blocking=True)
        evs[0].record(side)
    for i in range(K):
        with torch.cuda.stream(side):
            devbuf[(i + 1) % 4].copy_(pin[(i + 1) % 4], non_blocking=True)
            evs[(i + 1) % 4].record(side)
        torch.cuda.current_stream().wait_event(evs[i % 4])
        yield devbuf[i % 4]


def v_main_stream_copy():               # pinned -> device on the compute stream itself
    for i in range(K):
        devbuf[i % 4].copy_(pin[i % 4], non_blocking=True)
        yield devbuf[i % 4]


def v_thread_idle():                    # a second Python thread that only sleeps
    stop = threading.Event()
    th = threading.Thread(target=lambda: [time.sleep(0.001) for _ in iter(lambda: stop.is_set(), True)], daemon=True)
    th.start()
    for i in range(K):
        yield xd[i % 4]
    stop.set()
    th.join()


def v_thread_memcpy():                  # a second thread doing pageable -> pinned copies only (no GPU work)
    stop = threading.Event()

    def work():
        while not stop.is_set():
            pin[0].copy_(xs[1])
            time.sleep(0.005)
    th = threading.Thread(target=work, daemon=True)
    th.start()
    for i in range(K):
        yield xd[i % 4]
    stop.set()
    th.join()


def v_thread_h2d_main_wait():           # a second thread issuing side-stream H2D copies, consumer ignores them
    stop = threading.Event()

    def work():
        torch.cuda.set_device(dev)
        while not stop.is_set():
            with torch.cuda.stream(side):
                devbuf[3].copy_(pin[3], non_blocking=True)
            time.sleep(0.01)
    th = threading.Thread(target=work, daemon=True)
    th.start()
    for i in range(K):
        yield xd[i % 4]
    stop.set()
    th.join()


for name, gen in (("side stream, no thread", v_side_stream_no_thread), ("main-stream pinned copy", v_main_stream_copy),
                  ("idle 2nd thread", v_thread_idle), ("2nd thread host memcpy", v_thread_memcpy),
                  ("2nd thread side-stream H2D", v_thread_h2d_main_wait)):
    timed(name, gen())
    timed(name, gen())
