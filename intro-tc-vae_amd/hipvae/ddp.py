"""Data-parallel context: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl"
is RCCL on ROCm; "gloo" is used by the CPU tests of this logic).

The reference has no distributed code.  What a batch-sharded run needs to stay equal to the
full-batch reference step is exactly:
  1. gradient averaging of the half being trained (one flat all-reduce per phase);
  2. the means of the WHOLE global batch for the TC estimator (all-gather of mu [B_loc,D] ->
     [B_glob,D]; its adjoint is a reduce-scatter of dmu) -- logvar stays local because of the
     reference's transposed-variance indexing (ops.py:81);
  3. full-batch BatchNorm statistics (Sync-BN: all-reduce of per-channel fp64 moments).
Everything else (reconstruction, KL, exp-ELBO terms) is per-sample.
"""
import torch
import torch.distributed as dist
from torch.autograd import Function

_ctx = None


class Context:
    def __init__(self, group=None, sync_bn=True, force=False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sync_bn = sync_bn
        self.active = self.world > 1 or force


def init(group=None, sync_bn=True, force=False):
    """Activate data-parallel behaviour for every solver / BatchNorm in this process.  ``force`` keeps the
    collectives on for a group of ONE rank (a one-GPU rig exercising the RCCL branches; results are unchanged)."""
    global _ctx
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    _ctx = Context(group if group is not None else dist.group.WORLD, sync_bn, force)
    try:
        import models
        models.HipBatchNorm2d.sync_group = _ctx.group if (sync_bn and _ctx.active) else None
    except ImportError:  # models not importable in pure-host tests of this module
        pass
    return _ctx


def shutdown():
    global _ctx
    _ctx = None
    try:
        import models
        models.HipBatchNorm2d.sync_group = None
    except ImportError:
        pass


def get():
    return _ctx if (_ctx is not None and _ctx.active) else None


class _AllGatherRows(Function):
    """[B_loc, D] -> [B_glob, D] (rank-major); backward = reduce-scatter(sum) of the gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if dist.get_backend(group) == "gloo":
            dist.all_gather(list(out.chunk(world, dim=0)), x.contiguous(), group=group)
        else:
            dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        world, rank = dist.get_world_size(ctx.group), dist.get_rank(ctx.group)
        g = g.contiguous()
        rows = g.shape[0] // world
        if dist.get_backend(ctx.group) == "gloo":      # gloo has no reduce_scatter
            dist.all_reduce(g, group=ctx.group)
            return g[rank * rows:(rank + 1) * rows].clone(), None
        out = torch.empty((rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, group=ctx.group)
        return out, None


def all_gather_rows(x):
    c = get()
    return x if c is None else _AllGatherRows.apply(x, c.group)


def row_offset(local_rows):
    c = get()
    return 0 if c is None else c.rank * local_rows


def average_(flat):
    """In-place mean over ranks of a flat gradient buffer (one collective per model half)."""
    c = get()
    if c is None:
        return
    if dist.get_backend(c.group) == "gloo":
        dist.all_reduce(flat, group=c.group)
        flat.mul_(1.0 / c.world)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=c.group)


def average_async(flat):
    """average_ started asynchronously on the collective's own stream (RCCL): returns ``finish()``, which orders the
    current stream after the collective.  Work that does not touch ``flat`` may be issued in between (the intro
    solvers run the next phase's decoder-only forwards there).  gloo and single-process runs finish immediately."""
    c = get()
    if c is None:
        return lambda: None
    if dist.get_backend(c.group) == "gloo":
        average_(flat)
        return lambda: None
    work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=c.group, async_op=True)
    return work.wait


def graph_capturable():
    """Whether a data-parallel step may be captured into a hipGraph: RCCL collectives are stream-ordered kernels and
    capture like any other launch (every rank captures at the same step and replays in lock-step); gloo stages
    through the host and cannot.  Opt-in (``ITCV_DDP_GRAPH=1``) until it has been timed on a multi-GPU node: the
    one-GPU development box only reaches a group of one rank (tests/test_ddp.py)."""
    import os
    c = get()
    # no _wait_for_pending_works (another torch build): no way to wait for the watchdog's list to empty -> stay eager
    return (c is not None and dist.get_backend(c.group) == "nccl" and hasattr(c.group, "_wait_for_pending_works")
            and os.environ.get("ITCV_DDP_GRAPH", "0") == "1")


def drain_pending_collectives():
    """Block until RCCL's watchdog thread has retired every collective this process has issued so far (call after a
    device synchronize).  Needed in front of a hipGraph capture of the data-parallel step: see solvers/vae.py.
    ``ProcessGroup._wait_for_pending_works`` = c10d::ProcessGroupNCCL::waitForPendingWorks, which returns once the
    watchdog's work list is empty; gloo groups have no watchdog and return at once."""
    c = get()
    if c is None:
        return
    groups = {id(c.group): c.group}
    try:
        import models
        g = models.HipBatchNorm2d.sync_group
        if g is not None:
            groups[id(g)] = g
    except ImportError:
        pass
    for g in groups.values():
        if dist.get_backend(g) == "nccl":
            g._wait_for_pending_works()


def mean_scalars_(vec):
    """Average a small vector of per-rank loss scalars so every rank reports the global value."""
    c = get()
    if c is None:
        return
    dist.all_reduce(vec, group=c.group)
    vec.mul_(1.0 / c.world)
