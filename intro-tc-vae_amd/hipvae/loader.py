"""Input pipeline for the step: a ``WrappedDataLoader``-compatible wrapper
(/root/reference/dataset.py:16-27, used at train.py:146-159 with ``batch_to_device``) that keeps the GPU fed.

The reference's wrapper calls ``x.to(device)`` on a pageable host tensor inside the training loop: a synchronous,
staged copy in front of every step.  At a step time of ~20 ms that copy (and the DataLoader's collate before it) is
exposed.  ``PrefetchLoader`` instead runs a staging thread that

  * pulls batches from the wrapped loader and copies each into a ring of PINNED host buffers (allocated once per
    shape),
  * copies it host -> device on a side HIP stream into a ring of device buffers while the previous step runs (or its
    hipGraph replays; ``train_step`` blocks in its one read-back with the GIL released, which is when the thread works),
  * and hands the consumer a device tensor that is already resident: the only work left on the compute stream is one
    event wait.

Same surface as the reference class: ``PrefetchLoader(data_loader, pre_process)``, ``len()``, iteration yields
``pre_process(*batch)``; ``pre_process`` receives device tensors (the reference's ``batch_to_device`` then finds its
``.to(device)`` calls to be no-ops).  Everything that is not a tensor passes through untouched.  A slot of the ring is
reused only after the consumer has come back for the batch after it, i.e. after all work on that batch has been issued
to the compute stream (the reuse then waits for that work on the device): consume each batch within its iteration.

``flip_p`` > 0 moves the reference's only augmentation, ``transforms.RandomHorizontalFlip(p)`` (dataset.py:219-224, run
per sample by a PIL worker there), behind the copy: every 4-D float image batch is mirrored per sample with probability
``flip_p`` by one HIP kernel on the side stream (``itcv_hflip``; the coin flips come from a device generator seeded with
``flip_seed``).  The wrapped dataset then skips its own flip and the PIL workers only decode and resize.
"""
import ctypes
import queue
import threading

import torch


def _host_copy(dst, src):
    """pageable -> pinned as ONE plain memcpy with the GIL released.  ``Tensor.copy_`` would fan a 3 MB copy out over
    the intra-op (OpenMP) pool, whose workers then spin: measured on the 16-core share of the GPU box a staging thread
    doing that slowed the concurrent training step from 23 to 88 ms."""
    src = src.contiguous()
    ctypes.memmove(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size())


class _Slot:
    __slots__ = ("host", "dev", "aug", "ready", "free", "used")

    def __init__(self):
        self.host, self.dev, self.aug = [], [], []
        self.ready, self.free, self.used = torch.cuda.Event(), torch.cuda.Event(), False


class _Stop(Exception):
    pass


class PrefetchLoader:
    def __init__(self, data_loader, pre_process=None, device=None, depth=3, flip_p=0.0, flip_seed=0):
        """``depth``: ring slots (>= 2): one being consumed, the others landed or in flight ahead of it."""
        if depth < 2:
            raise ValueError("depth must be >= 2")
        self.dl = data_loader
        self.func = pre_process
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise ValueError("PrefetchLoader feeds a HIP device; use the reference's WrappedDataLoader on the CPU")
        self.depth = depth
        self.flip_p = float(flip_p)
        self._flip_gen = None
        self._flip_seed = int(flip_seed)
        self._stream = None
        self._slots = None

    def __len__(self):
        return len(self.dl)

    # ---- staging thread: one batch -> slot (pinned copy + async H2D on the side stream) -------------------
    def _stage(self, slot, batch):
        items = batch if isinstance(batch, (tuple, list)) else (batch,)
        out = []
        side = self._stream
        if slot.used:
            side.wait_event(slot.free)          # device side: the step that consumed this slot's previous batch has run
            slot.ready.synchronize()            # host side: the pinned buffers' previous copies have left the host
        ti = 0
        for it in items:
            if not isinstance(it, torch.Tensor) or it.device.type == "cuda":
                out.append(it)                  # non-tensors, and tensors already resident (a GPU-side dataset)
                continue
            if ti == len(slot.host) or slot.host[ti].shape != it.shape or slot.host[ti].dtype != it.dtype:
                host = torch.empty(it.shape, dtype=it.dtype, pin_memory=True)
                dev = torch.empty(it.shape, dtype=it.dtype, device=self.device)
                if ti == len(slot.host):
                    slot.host.append(host), slot.dev.append(dev)
                else:
                    slot.host[ti], slot.dev[ti] = host, dev
            _host_copy(slot.host[ti], it)       # pageable -> pinned (the DataLoader's tensors are pageable)
            with torch.cuda.stream(side):
                slot.dev[ti].copy_(slot.host[ti], non_blocking=True)
                staged = slot.dev[ti]
                if self.flip_p > 0.0 and it.dim() == 4 and it.dtype == torch.float32:
                    staged = self._flip(slot, ti, staged, side)
            out.append(staged)
            ti += 1
        slot.ready.record(side)
        slot.used = True
        return out if isinstance(batch, (tuple, list)) else out[0]

    def _flip(self, slot, ti, x, side):
        """RandomHorizontalFlip(p) of an image batch on the side stream (already current): per-sample coin flips from the
        device generator, one kernel into the slot's second device buffer."""
        from . import abi
        if self._flip_gen is None:
            self._flip_gen = torch.Generator(device=self.device)
            self._flip_gen.manual_seed(self._flip_seed)
        while len(slot.aug) <= ti:
            slot.aug.append(None)
        if slot.aug[ti] is None or slot.aug[ti].shape != x.shape:
            slot.aug[ti] = torch.empty_like(x)
        B, C, H, W = x.shape
        coins = (torch.rand(B, device=self.device, generator=self._flip_gen) < self.flip_p).to(torch.uint8)
        abi.call("itcv_hflip", abi.ptr(x), abi.ptr(slot.aug[ti]), coins.data_ptr(), B, C * H, W, side.cuda_stream)
        coins.record_stream(side)
        return slot.aug[ti]

    def _producer(self, q, credits, stop):
        try:
            torch.cuda.set_device(self.device)
            k = 0
            for b in self.dl:
                while not credits.acquire(timeout=0.1):     # a free ring slot (released by the consumer)
                    if stop.is_set():
                        raise _Stop
                if stop.is_set():
                    raise _Stop
                slot = self._slots[k % self.depth]
                k += 1
                q.put((slot, self._stage(slot, b)))
            q.put(None)
        except _Stop:
            pass
        except BaseException as e:  # noqa: BLE001 -- re-raised in the consumer
            q.put(e)

    def __iter__(self):
        # one producer at a time: an iteration abandoned early (break / exception) has stopped its producer in `finally`
        # below (joined without a timeout); a second iterator started WHILE one is still live would share the ring
        old = getattr(self, "_thread", None)
        if old is not None and old.is_alive():
            raise RuntimeError("PrefetchLoader: the previous iteration is still running (one iterator at a time)")
        if self._stream is None:
            with torch.cuda.device(self.device):
                self._stream = torch.cuda.Stream(device=self.device)
        # a fresh ring per iteration: no slot state (used flags, events) survives an abandoned epoch
        self._slots = [_Slot() for _ in range(self.depth)]
        q, credits, stop = queue.Queue(), threading.Semaphore(self.depth), threading.Event()
        th = self._thread = threading.Thread(target=self._producer, args=(q, credits, stop), daemon=True, name="itcv-prefetch")
        th.start()
        prev = None
        try:
            while True:
                item = q.get()
                cur = torch.cuda.current_stream(self.device)
                if prev is not None:
                    # everything the consumer issued for the previous batch is on `cur` by now: its slot may be reused
                    # once that work has run
                    prev.free.record(cur)
                    credits.release()
                    prev = None
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, staged = item
                cur.wait_event(slot.ready)
                prev = slot
                if self.func is None:
                    yield staged
                elif isinstance(staged, (tuple, list)):
                    yield self.func(*staged)
                else:
                    yield self.func(staged)
        finally:
            stop.set()
            if prev is not None:
                prev.free.record(torch.cuda.current_stream(self.device))
            # the producer polls `stop` every 0.1 s while it waits for a slot and checks it after every batch of the wrapped
            # loader: it ends on its own; wait for it (a producer still inside the wrapped loader must not outlive this iterator)
            th.join()
            torch.cuda.current_stream(self.device).wait_stream(self._stream)   # copies still in flight land before the ring is dropped
