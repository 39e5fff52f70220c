"""torch.autograd.Function wrappers around the HIP kernels of libitcv_hip.so.

Each Function forwards device pointers to one or a few C-ABI entry points (hipvae.abi) and
keeps only what its backward needs.  PyTorch provides memory, the stream and the autograd
tape; all arithmetic on activations, gradients and statistics happens in the HIP kernels.
The optional ``group`` arguments are torch.distributed process groups: they turn BatchNorm
into Sync-BN (all-reduce of the fp64 channel moments over RCCL) for data-parallel parity
with the full-batch reference.
"""
import contextlib
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import abi
from .abi import call, lib, ptr, stream

F32 = torch.float32


def set_option(name, value):
    """Launch-shape options of the library (include/itcv_hip.h: itcv_set_option): 'band_m16' (0/1),
    'band_persist_blocks' (0 = one tile per block, else the persistent kernel's block count), 'wgrad_m16' (0 / 1 =
    32x32x16 / 16x16x32 products in the planes weight gradient), 'planes_mfma_waves' (4 / 8).  Validated by the library."""
    call("itcv_set_option", name.encode(), int(value))


def get_option(name):
    return lib.itcv_get_option(name.encode())


@contextlib.contextmanager
def option_scope(name, value):
    prev = get_option(name)
    set_option(name, value)
    try:
        yield
    finally:
        set_option(name, prev)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _f32c(t):
    if t.dtype != F32:
        raise abi.HipExtensionError(f"HIP path is fp32 only (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------ gradient accumulation mode
_DIRECT = [False]


@contextlib.contextmanager
def direct_grad_accumulation():
    """Inside this block the backward kernels add parameter gradients straight into an existing,
    contiguous ``param.grad`` (the solvers' flat gradient buffers) and hand autograd ``None``,
    instead of returning fresh tensors for autograd to add (saves one elementwise pass and one
    launch per parameter per network pass)."""
    prev, _DIRECT[0] = _DIRECT[0], True
    try:
        yield
    finally:
        _DIRECT[0] = prev


def _grad_target(param):
    g = param.grad if _DIRECT[0] else None
    return g if (g is not None and g.is_contiguous() and g.dtype == F32) else None


# (Round 1 issued the weight-gradient GEMMs on a second HIP stream; with the batched passes of round 2 they fill the chip
# on their own and running them beside the data-gradient chain cost more than the gaps it filled -- same-box A/B of the c2
# step 18.45 vs 18.0 ms -- so the side stream is gone.)

# ------------------------------------------------------------------ convolution / linear
_WEIGHT_EPOCH = [0]
_PACK_CACHE = {}


_PARAM_EPOCH = {}


def bump_weight_epoch(params=None):
    """Call after parameters were modified through raw pointers (fused Adam): invalidates the packed-weight
    cache (torch's own in-place ops are tracked through ``tensor._version``).  ``params``: only these tensors
    changed (the optimiser step of one model half leaves the other half's packed weights valid)."""
    if params is None:
        _WEIGHT_EPOCH[0] += 1
        return
    for p in params:
        _PARAM_EPOCH[id(p)] = _PARAM_EPOCH.get(id(p), 0) + 1


# conv arithmetic -> plane format code of the C ABI (include/itcv_hip.h): 2 / 3 bf16 planes, 4 = two fp16 planes + scale
_NS = {"fp32": 0, "bf16x3": 2, "bf16x6": 3, "f16x3": 4}
F16X2 = 4
import os as _os

_CONV_MATH = [_os.environ.get("ITCV_CONV_MATH", "fp32")]   # the one documented environment override (with ITCV_DDP_GRAPH, ITCV_LIB)
assert _CONV_MATH[0] in _NS, "ITCV_CONV_MATH must be one of fp32 / bf16x3 / bf16x6 / f16x3"


def _two(ns):
    """Two-plane formats (bf16x3, f16x3): the ones the weight-gradient and 5x5 planes kernels take."""
    return ns in (2, F16X2)


# Module switches used by the test matrix (never read from the environment):
_SMALL_PLANES = [True]   # matrix-core form of the 3-output 5x5 convs (tests compare it with the direct fp32 kernels)
_POISON = [False]        # fill planes-only tensors with NaN (tests: nothing may read an fp32 tensor that was not written)
_PLANES = [True]         # split-bf16 convs take pre-split operands (tests: planes kernels == gather kernels, bit for bit)


def set_conv_math(mode):
    """Arithmetic of the conv/linear forward and data-gradient GEMMs:
    'fp32'   exact fp32 MFMA (v_mfma_f32_32x32x2_f32) -- the parity path and the default;
    'bf16x6' operands split into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 accumulate: fp32-class
             accuracy (~2^-23 per product) at 2.7x the matrix-core rate;
    'bf16x3' 2 planes / 3 MFMAs: ~2^-16 per product at 5.3x the matrix-core rate;
    'f16x3'  2 FP16 planes (hi, lo of the tensor times a power-of-two scale) / the same 3 MFMAs on the fp16 matrix
             cores: 22 significand bits, ~2^-21 per product -- fp32 class at the bf16x3 rate.  Shapes outside the planes
             kernels run on the exact fp32 kernels in this mode.
    Layers the split kernel does not cover (3-channel stem/predict, KS=5) stay on the fp32 kernel."""
    assert mode in _NS, mode
    _CONV_MATH[0] = mode


def conv_math():
    return _CONV_MATH[0]


@contextlib.contextmanager
def conv_math_scope(mode):
    """Temporarily select the conv arithmetic (used by the solvers: ``use_amp`` picks the mode)."""
    if mode is None:
        yield
        return
    assert mode in _NS, mode
    prev, _CONV_MATH[0] = _CONV_MATH[0], mode
    try:
        yield
    finally:
        _CONV_MATH[0] = prev


def _pack_key(weight):
    return (weight.data_ptr(), weight._version, _WEIGHT_EPOCH[0], _PARAM_EPOCH.get(id(weight), 0))


def _pack_entry(weight, key):
    ent = _PACK_CACHE.get(id(weight))
    if ent is None:
        ent = _PACK_CACHE[id(weight)] = [key, {}]
        weakref.finalize(weight, _PACK_CACHE.pop, id(weight), None)
        weakref.finalize(weight, _PARAM_EPOCH.pop, id(weight), None)
    elif ent[0] != key:
        ent[0], ent[1] = key, {}
    return ent


def packed_weight(weight, w4, for_dgrad, ns=0):
    """Packed operand of ``weight`` (viewed as ``w4`` [Co,Ci,KS,KS]), cached until the weight changes:
    the frozen half of the model is packed once per phase instead of once per network pass.  Weights registered
    through register_pack_group() (the parameters of one optimiser) are re-packed together, one launch per
    (direction, arithmetic), when the first of them is asked for after the optimiser step."""
    ent = _pack_entry(weight, _pack_key(weight))
    wp = ent[1].get((for_dgrad, ns))
    if wp is None:
        grp = _PACK_GROUPS.get(id(weight)) if ns and _PACK_BATCH[0] else None
        if grp is not None:
            wp = grp.pack(weight, w4, for_dgrad, ns)
        else:
            wp = ent[1][(for_dgrad, ns)] = pack_weight(w4, for_dgrad) if ns == 0 else pack_weight_bf16s(w4, for_dgrad, ns)
    return wp


# One launch re-packs every split-bf16 conv weight of a parameter group (False: one launch per layer; the tests compare the two).
_PACK_BATCH = [True]
_PACK_GROUPS = {}


class _PackGroup:
    """Conv weights that change together.  Membership per (for_dgrad, ns) is learnt from the requests: a weight seen for
    the first time is packed on its own and joins; from then on a miss on any member re-packs all of them into their
    persistent buffers through one device-resident descriptor table."""

    def __init__(self):
        self.members = {}    # (for_dgrad, ns) -> {id(weight): [weakref(weight), (co, ci, ks), data_ptr, wp]}
        self.tables = {}     # (for_dgrad, ns) -> (dev_table, n, total_blocks)

    def pack(self, weight, w4, for_dgrad, ns):
        k = (for_dgrad, ns)
        mem = self.members.setdefault(k, {})
        rec = mem.get(id(weight))
        if rec is None or rec[2] != w4.data_ptr() or rec[1] != tuple(w4.shape[:3]):
            wp = pack_weight_bf16s(w4, for_dgrad, ns)
            mem[id(weight)] = [weakref.ref(weight), tuple(w4.shape[:3]), w4.data_ptr(), wp]
            self.tables.pop(k, None)
            _pack_entry(weight, _pack_key(weight))[1][k] = wp
            return wp
        # The table holds every member's raw source pointer: before a grouped launch each member must still be alive and
        # sit where it was registered (a `p.data` reassignment or a freed weight would be re-packed from stale memory).
        # Members that moved or died leave the group (and drop the table); they re-join on their next own request.
        stale = [i for i, r in mem.items()
                 if r[0]() is None or r[0]().data_ptr() != r[2] or tuple(r[0]().shape[:3]) != r[1]]
        if stale:
            for i in stale:
                del mem[i]
            self.tables.pop(k, None)
        tab = self.tables.get(k)
        if tab is None:
            tab = self.tables[k] = self._build(mem, for_dgrad, ns, weight.device)
        call("itcv_conv2d_pack_weights_bf16s", ptr(tab[0]), tab[1], tab[2], ns, stream())
        for r in mem.values():        # every remaining member was just verified: its buffer holds the fresh packing
            m = r[0]()
            _pack_entry(m, _pack_key(m))[1][k] = r[3]
        return rec[3]

    @staticmethod
    def _build(mem, for_dgrad, ns, device):
        import ctypes

        nb = lib.itcv_pack_desc_bytes()
        host = (ctypes.c_uint8 * (nb * len(mem)))()
        blocks = 0
        for i, r in enumerate(mem.values()):
            co, ci, ks = r[1]
            got = lib.itcv_conv2d_pack_desc_bf16s(ctypes.byref(host, i * nb), r[2], ptr(r[3]), co, ci, ks, int(for_dgrad), ns,
                                                  blocks)
            if got <= 0:
                raise abi.HipExtensionError("itcv_conv2d_pack_desc_bf16s: " + abi.last_error())
            blocks += got
        dev_table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
        return dev_table, len(mem), blocks


def register_pack_group(params):
    """Parameters that one optimiser step changes together (hipvae.flat.FlatGroup): their packed conv operands are
    refreshed by one launch per direction."""
    grp = _PackGroup()
    for p in params:
        if p.dim() == 4:
            _PACK_GROUPS[id(p)] = grp
            weakref.finalize(p, _PACK_GROUPS.pop, id(p), None)
    return grp


def pack_weight_bf16s(w4, for_dgrad, ns):
    co, ci, ks = w4.shape[0], w4.shape[1], w4.shape[2]
    nbytes = lib.itcv_conv2d_packed_weight_bytes_bf16s(co, ci, ks, int(for_dgrad), ns)
    wp = torch.empty(nbytes // 4, dtype=torch.int32, device=w4.device)
    call("itcv_conv2d_pack_weight_bf16s", ptr(w4), ptr(wp), co, ci, ks, int(for_dgrad), ns, stream())
    return wp


# bf16x3 mode: the 3 -> 64 stem conv and the prediction layer's data-gradient on the matrix cores (False: the direct fp32
# kernel, as in the other arithmetic modes; the tests compare the two)
_SCIN_MFMA = [True]


def conv_apply(x, weight, w4, for_dgrad, bias, B, Ci, H, W, Co, KS, up2):
    """Forward-type conv GEMM (forward, or data-gradient with the roles of Ci/Co swapped by the caller)
    on the kernel selected by set_conv_math()."""
    if not up2 and lib.itcv_conv2d_small_cout_supported(Co, KS):
        # <= 4 output channels: direct fp32 conv on the vector ALUs, reads the raw OIHW weights
        y = torch.empty((B, Co, H, W), dtype=F32, device=x.device)

        call("itcv_conv2d_small_cout_fwd", ptr(x), ptr(w4), ptr(bias), ptr(y), B, Ci, H, W, Co, KS, int(for_dgrad),
             stream())
        return y
    if not up2 and lib.itcv_conv2d_small_cin_supported(Ci, KS):
        # <= 4 reduction channels: direct fp32 conv, the pixel's input window lives in registers
        y = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
        fmt = _NS[_CONV_MATH[0]]
        if _SCIN_MFMA[0] and _two(fmt) and lib.itcv_conv2d_small_cin_bf16x3_supported(Ci, Co, KS, W):
            # fp16 form: the data-gradient's input is a gradient tensor -> scale from its magnitude (device side)
            amax = absmax_parts(x) if (fmt == F16X2 and for_dgrad) else None
            call("itcv_conv2d_small_cin_fwd_bf16x3", ptr(x), ptr(w4), ptr(bias), ptr(y), B, Ci, H, W, Co, KS,
                 int(for_dgrad), fmt, ptr(amax), stream())
            return y

        call("itcv_conv2d_small_cin_fwd", ptr(x), ptr(w4), ptr(bias), ptr(y), B, Ci, H, W, Co, KS, int(for_dgrad),
             stream())
        return y
    ns = _NS[_CONV_MATH[0]]
    if ns in (2, 3) and lib.itcv_conv2d_bf16s_supported(Ci, Co, KS):
        wp = packed_weight(weight, w4, for_dgrad, ns)
        y = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
        nws = lib.itcv_conv2d_fwd_bf16s_workspace(B, Ci, H, W, Co, KS)
        ws = _ws(nws, x.device) if nws else None

        call("itcv_conv2d_fwd_bf16s", ptr(x), ptr(wp), ptr(bias), ptr(y), B, Ci, H, W, Co, KS, int(up2), ns, ptr(ws), nws,
             stream())
        return y
    return conv_fwd_raw(x, packed_weight(weight, w4, for_dgrad), bias, B, Ci, H, W, Co, KS, up2)


def absmax_parts(x):
    """256 block maxima of |x| (device side): what the fp16 split kernels derive a gradient tensor's scale from."""
    parts = torch.empty(256, dtype=F32, device=x.device)
    call("itcv_absmax", ptr(x), x.numel(), ptr(parts), stream())
    return parts


def split_planes(x, ns, gradient=False):
    """fp32 [B,C,H,W] -> pre-split planes [planes][B][C/8][H][W] x 16 B in format ``ns`` (see include/itcv_hip.h).
    fp16 planes (ns = 4): activations are split with scale 1; a ``gradient`` tensor (arbitrary magnitude) with the
    power-of-two scale that maps its largest element just under 2^15 (two launches: maxima, split)."""
    B, C, H, W = x.shape
    nbytes = lib.itcv_planes_bytes(B, C, H * W, ns)
    if not nbytes:
        raise abi.HipExtensionError(f"split_planes: unsupported shape {tuple(x.shape)} / ns={ns}")
    xp = torch.empty(nbytes // 4, dtype=torch.int32, device=x.device)
    if ns == F16X2 and gradient:
        call("itcv_split_planes_scaled", ptr(x), ptr(xp), B, C, H * W, ns, ptr(absmax_parts(x)), stream())
    else:
        call("itcv_split_planes", ptr(x), ptr(xp), B, C, H * W, ns, stream())
    return xp


# BatchNorm statistics from the conv epilogue (itcv_conv2d_fwd_bf16p_st).  OFF by default: measured, the staged epilogue
# that produces them costs the band kernel more (+8 % on the 64-channel layers) than the statistics pass it replaces
# saves -- that pass reads the conv output out of the Infinity Cache right behind the conv (4-10 us) -- see DESIGN.md.
# Kept as a tested option of the C ABI (itcv_conv2d_fwd_bf16p_st), not reachable from the environment.
_FUSE_STATS = [False]


def conv_apply_planes(xp, weight, w4, for_dgrad, bias, B, Ci, H, W, Co, KS, up2, ns, want_stats=False):
    """conv_apply with the input given as pre-split planes (LDS-DMA kernel, no gather).  ``want_stats``: where the
    kernel can, it also leaves the per-tile channel sums of its output for the BatchNorm that follows
    (attached to the result as ``_itcv_tile_stats``; BnActFn then skips its own statistics pass)."""
    if not up2 and _two(ns) and lib.itcv_conv2d_small_cout_bf16p_supported(Ci, Co, KS):
        y = torch.empty((B, Co, H, W), dtype=F32, device=xp.device)
        call("itcv_conv2d_small_cout_fwd_bf16p", ptr(xp), ptr(w4), ptr(bias), ptr(y), B, Ci, H, W, Co, KS, int(for_dgrad),
             ns, stream())
        return y
    wp = packed_weight(weight, w4, for_dgrad, ns)
    y = torch.empty((B, Co, H, W), dtype=F32, device=xp.device)
    nws = lib.itcv_conv2d_fwd_bf16p_workspace(B, Ci, H, W, Co, KS, ns)
    ws = _ws(nws, xp.device) if nws else None
    T = lib.itcv_conv2d_fwd_bf16p_stat_tiles(B, Ci, H, W, Co, KS, ns) if (want_stats and _FUSE_STATS[0]) else 0
    stats = torch.empty((2, Co, T), dtype=F32, device=xp.device) if T else None
    call("itcv_conv2d_fwd_bf16p_st", ptr(xp), ptr(wp), ptr(bias), ptr(y), B, Ci, H, W, Co, KS, int(up2), ns, ptr(stats),
         ptr(ws), nws, stream())
    if stats is not None:
        y._itcv_tile_stats = (stats, T, 256, y._version, y.data_ptr())     # 256 = (image, pixel) positions per tile
    return y


def _tile_stats_of(x, B, G, HW):
    """(stats, tiles per group, pitch) when ``x`` carries the producing conv's per-tile channel sums and the BatchNorm
    groups are whole numbers of tiles, else None."""
    tag = getattr(x, "_itcv_tile_stats", None)
    if tag is None or tag[3] != x._version or tag[4] != x.data_ptr():
        return None
    stats, T, px = tag[0], tag[1], tag[2]
    if T * px != B * HW or (B // G * HW) % px:
        return None
    return stats, T // G, T


# ---- deferred slab reduces of the planes weight gradients ------------------------------------------------------------
# Inside deferred_wgrad_reduces() (the solvers wrap loss.backward() in it) a weight gradient that accumulates into an
# existing .grad leaves its split-K slabs in a persistent buffer and is folded, together with all the other layers of the
# backward pass, by ONE table-driven launch when the block ends (36 reduce launches per intro step -> 3).  Same summation
# order as the per-call reduce: bitwise equal.  Slab buffers are kept per position in the backward's call sequence and the
# device tables per sequence of (slab, target) pointers, so a steady-state step (and its hipGraph) allocates and copies
# nothing.
_DEFER = {"on": False, "pending": [], "pool": {}, "tables": {}}


@contextlib.contextmanager
def deferred_wgrad_reduces():
    prev, _DEFER["on"] = _DEFER["on"], True
    try:
        yield
    finally:
        _DEFER["on"] = prev
        if not prev:
            flush_wgrad_reduces()


def flush_wgrad_reduces():
    pend = _DEFER["pending"]
    if not pend:
        return
    import ctypes
    # group the calls by target (a weight used by several network passes of the backward), keeping call order
    groups = {}
    for ws, dw, co, ci, slabs in pend:
        groups.setdefault(dw.data_ptr(), [dw, co, ci, []])[3].append((ws, slabs))
    key = tuple((k, tuple((w.data_ptr(), n) for w, n in g[3])) for k, g in groups.items())
    tab = _DEFER["tables"].get(key)
    if tab is None:
        nb = lib.itcv_wgrad_reduce_desc_bytes()
        descs, blocks = [], 0
        for dw, co, ci, srcs in groups.values():
            for c0 in range(0, len(srcs), 4):          # at most four slab sources per descriptor; later ones accumulate
                part = srcs[c0:c0 + 4]
                descs.append((dw, co, ci, part, 1))
        host = (ctypes.c_uint8 * (nb * len(descs)))()
        for i, (dw, co, ci, part, acc) in enumerate(descs):
            sl = (ctypes.c_void_p * len(part))(*[w.data_ptr() for w, _ in part])
            sp = (ctypes.c_int * len(part))(*[n for _, n in part])
            got = lib.itcv_wgrad_reduce_desc(ctypes.byref(host, i * nb), sl, sp, len(part), dw.data_ptr(), co, ci, acc, blocks)
            if got <= 0:
                raise abi.HipExtensionError("itcv_wgrad_reduce_desc: " + abi.last_error())
            blocks += got
        # descriptors of one target that were split into several (> 4 sources) would race inside one launch
        if len(descs) != len(groups):
            raise abi.HipExtensionError("deferred weight-gradient reduce: more than four passes over one weight")
        if len(_DEFER["tables"]) > 64:
            _DEFER["tables"].clear()
        tab = _DEFER["tables"][key] = (torch.frombuffer(bytearray(host), dtype=torch.uint8).to(pend[0][1].device), len(descs), blocks)
    call("itcv_wgrad_reduce_many", ptr(tab[0]), tab[1], tab[2], stream())
    pend.clear()


def _defer_slab_buffer(nbytes, device):
    """Persistent slab buffer for the next deferred call: one per position in the pending sequence (and size)."""
    k = (len(_DEFER["pending"]), int(nbytes), str(device))
    buf = _DEFER["pool"].get(k)
    if buf is None:
        buf = _DEFER["pool"][k] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return buf


def conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, KS, up2, out=None, accumulate=False, ns=2):
    """Weight gradient from the pre-split planes of x and dy (two-plane formats; transposing-LDS-read kernel)."""
    dw = out if out is not None else torch.empty((Co, Ci, KS, KS), dtype=F32, device=xp.device)
    nws = lib.itcv_conv2d_wgrad_bf16p_workspace(B, Ci, H, W, Co, KS)
    if _DEFER["on"] and out is not None and accumulate:
        same = sum(1 for p in _DEFER["pending"] if p[1].data_ptr() == dw.data_ptr())
        if same >= 4 or (same == 0 and len({p[1].data_ptr() for p in _DEFER["pending"]}) >= lib.itcv_wgrad_reduce_max_descs()):
            flush_wgrad_reduces()                      # four sources per weight, a bounded table per launch
        ws = _defer_slab_buffer(nws, xp.device)
        call("itcv_conv2d_wgrad_bf16p", ptr(xp), ptr(dyp), ptr(dw), B, Ci, H, W, Co, KS, int(up2), int(ns), 2, ptr(ws), nws,
             stream())
        _DEFER["pending"].append((ws, dw, Co, Ci, lib.itcv_conv2d_wgrad_bf16p_slabs(B, Ci, H, W, Co, KS)))
        # (flushing mid-backward, every 48 / 96 / 192 MB of slabs, so that the fold reads them out of the Infinity Cache,
        # measured slower on one box: 16.90-16.93 ms per c2 step against 16.70-16.78 with one fold per backward)
        return dw
    ws = _ws(nws, xp.device)

    call("itcv_conv2d_wgrad_bf16p", ptr(xp), ptr(dyp), ptr(dw), B, Ci, H, W, Co, KS, int(up2), int(ns), int(accumulate),
         ptr(ws), nws, stream())
    return dw


def pack_weight(w4, for_dgrad):
    """w4 [Co,Ci,KS,KS] -> packed K-major operand (see include/itcv_hip.h)."""
    co, ci, ks = w4.shape[0], w4.shape[1], w4.shape[2]
    wp = torch.empty(lib.itcv_conv2d_packed_weight_elems(co, ci, ks, for_dgrad), dtype=F32, device=w4.device)
    call("itcv_conv2d_pack_weight", ptr(w4), ptr(wp), co, ci, ks, int(for_dgrad), stream())
    return wp


class LaunchProfile:
    """Per-launch timing of the GEMM-class kernels, recorded INSIDE libitcv_hip.so: a HIP event pair on
    the launch stream around the main kernel of every conv call (bench.py's roofline leg)."""
    KINDS = {0: "conv_fwd_kernel", 1: "conv_fwd_bf16s_kernel", 2: "conv_wgrad_kernel", 3: "conv_wgrad_bf16s_kernel",
             4: "conv_small_cout_kernel", 5: "conv_small_cin_kernel", 6: "conv_fwd_bf16p_kernel",
             7: "conv_wgrad_bf16p_kernel", 8: "conv_fwd_bf16p2_kernel", 9: "conv_fwd_bf16p3_kernel",
             10: "conv_small_cout_planes_kernel", 11: "conv_small_cin_mfma_kernel", 12: "conv_wgrad5_planes_kernel",
             13: "bn_act_fwd_planes_kernel", 14: "bn_bwd_apply_planes"}
    HBM_KINDS = (13, 14)     # HBM-bound kernels: the record's "work" is algorithmic BYTES, not FLOP

    @classmethod
    def begin(cls):
        call("itcv_profile_begin")

    @classmethod
    def end(cls):
        """-> list of (kernel label, algorithmic FLOP, seconds)."""
        import ctypes
        n = lib.itcv_profile_end()
        code, flop, ms = ctypes.c_int(), ctypes.c_double(), ctypes.c_float()
        out = []
        for i in range(n):
            call("itcv_profile_get", i, ctypes.byref(code), ctypes.byref(flop), ctypes.byref(ms))
            c = code.value
            kind, ks, bm, up2, ns = c & 15, (c >> 4) & 15, (c >> 8) & 255, (c >> 16) & 1, (c >> 20) & 15
            if kind in (0, 2):
                label = f"{cls.KINDS[kind]}<KS={ks},BM={bm},up2={up2}>"
            elif kind in (7, 8, 9):   # templates <LOG2W, ...>: the KS field carries log2(W); 9 = persistent band kernel
                label = f"{cls.KINDS[kind]}<LOG2W={ks},BM={bm},up2={up2},NS={ns}>"
            elif kind in (1, 3, 6):
                label = f"{cls.KINDS[kind]}<KS={ks},BM={bm},up2={up2},NS={ns}>"
            elif kind in cls.HBM_KINDS:    # BatchNorm apply passes: channels, image width, pool / adjoint mode, plane format
                label = f"{cls.KINDS[kind]}<C={8 * bm},W={1 << ks},mode={(c >> 16) & 15},NS={ns}>"
            elif kind in (10, 11, 12):     # the 5x5 layers on the matrix cores: C = the narrow side's channels
                label = f"{cls.KINDS[kind]}<KS={ks},C={bm},stem={up2},NS={ns}>"
            else:
                label = f"{cls.KINDS[kind]}<KS={ks},C={bm}>"
            out.append((label, flop.value, ms.value * 1e-3))
        call("itcv_profile_clear")
        return out


def conv_fwd_raw(x, wp, bias, B, Ci, H, W, Co, KS, up2):
    y = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
    nws = lib.itcv_conv2d_fwd_workspace(B, Ci, H, W, Co, KS)
    ws = _ws(nws, x.device) if nws else None

    call("itcv_conv2d_fwd", ptr(x), ptr(wp), ptr(bias), ptr(y), B, Ci, H, W, Co, KS, int(up2), ptr(ws), nws, stream())
    return y


def conv_wgrad_raw(x, dy, B, Ci, H, W, Co, KS, up2, out=None, accumulate=False):
    dw = out if out is not None else torch.empty((Co, Ci, KS, KS), dtype=F32, device=x.device)
    nws = lib.itcv_conv2d_wgrad_workspace(B, Ci, H, W, Co, KS)
    ws = _ws(nws, x.device)
    ns = _NS[_CONV_MATH[0]]
    if ns in (2, 3) and lib.itcv_conv2d_wgrad_bf16s_supported(Ci, H, W, Co, KS):
        if up2:   # the split kernel reads 8-pixel chunks: materialise the x2 nearest upsampling once
            xu = torch.empty((B, Ci, H, W), dtype=F32, device=x.device)
            call("itcv_upsample2_fwd", ptr(x), ptr(xu), B * Ci, H // 2, W // 2, stream())
            x = xu

        call("itcv_conv2d_wgrad_bf16s", ptr(x), ptr(dy), ptr(dw), B, Ci, H, W, Co, KS, ns, int(accumulate), ptr(ws), nws,
             stream())
        return dw

    call("itcv_conv2d_wgrad", ptr(x), ptr(dy), ptr(dw), B, Ci, H, W, Co, KS, int(up2), int(accumulate), ptr(ws), nws,
         stream())
    return dw


def bias_grad_raw(dy, B, C, HW, target=None):
    """db = sum_{b,hw} dy; with ``target`` the sum is added into it and None is returned."""
    db = target if target is not None else torch.empty((C,), dtype=F32, device=dy.device)
    nws = lib.itcv_bias_grad_workspace(B, C, HW)
    ws = _ws(nws, dy.device)
    call("itcv_bias_grad", ptr(dy), ptr(db), B, C, HW, int(target is not None), ptr(ws), nws, stream())
    return None if target is not None else db


PLANES_STATS = [0, 0]   # conv operands taken from producer-attached planes / split on demand (diagnostic)


def _tag_planes(t, planes, ns, fp32_valid=True):
    """Attach the pre-split planes of ``t`` to the tensor object (consumed by the next conv GEMM).
    ``fp32_valid=False``: the producer did not write ``t`` itself -- only the planes carry its values."""
    t._itcv_planes = (planes, ns, t._version, t.data_ptr(), fp32_valid)


def _require_fp32(t, what):
    tag = getattr(t, "_itcv_planes", None)
    if tag is not None and not tag[4]:
        raise abi.HipExtensionError(f"{what}: this tensor was produced as planes only (fp32 values not written)")
    return t


def _tagged_planes(t, ns):
    tag = getattr(t, "_itcv_planes", None)
    if tag is None or tag[1] != ns or tag[2] != t._version or tag[3] != t.data_ptr():
        return None
    return tag[0]


def planes_of(t, ns, gradient=False):
    """Planes of ``t``: the ones its producer attached, else a split pass (``gradient``: see split_planes)."""
    p = _tagged_planes(t, ns)
    PLANES_STATS[0 if p is not None else 1] += 1
    return p if p is not None else split_planes(t, ns, gradient)


def conv_input_planes_ns(conv, up2=False):
    """ns when ``conv`` (an nn.Conv2d-like module) will read its input as planes in the current mode, else 0:
    tells the producing BatchNorm pass to emit them."""
    if conv is None:
        return 0
    return _planes_ns(conv.in_channels, conv.out_channels, conv.kernel_size[0], up2)


def conv_grad_planes_ns(conv, needs_input_grad=True):
    """ns when the backward of ``conv`` will read its output gradient as planes (data- and/or weight-gradient)."""
    if conv is None:
        return 0
    ks = conv.kernel_size[0]
    ns = _planes_ns(conv.out_channels, conv.in_channels, ks, False) if needs_input_grad else 0
    if not ns and _two(_NS[_CONV_MATH[0]]) and _PLANES[0] and (ks == 3 or (ks == 5 and _SMALL_PLANES[0]
                                                                           and conv.in_channels <= 3
                                                                           and conv.out_channels == 64)):
        ns = _NS[_CONV_MATH[0]]    # weight gradient on planes (shape support is re-checked where the planes are consumed)
    return ns


def conv_input_mode(conv, B, H, W, up2=False):
    """(ns, fp32_needed) for the tensor feeding ``conv`` whose OUTPUT is [B, Co, H, W]: the number of planes its
    producer should emit, and whether the fp32 tensor itself is still read (fallback kernels)."""
    ns = conv_input_planes_ns(conv, up2)
    if not ns:
        return 0, True
    ks = conv.kernel_size[0]
    wg_ok = (_wgrad_planes_ok(B, conv.in_channels, H, W, conv.out_channels, ks)
             or _wgrad5_mode(conv.in_channels, H, W, conv.out_channels, ks, up2) == "predict")
    return ns, not (_two(ns) and wg_ok)


def conv_grad_mode(conv, B, H, W, needs_input_grad=True):
    """(ns, fp32_needed) for the gradient of ``conv``'s output [B, Co, H, W]."""
    ns = conv_grad_planes_ns(conv, needs_input_grad)
    if not ns:
        return 0, True
    ks = conv.kernel_size[0]
    dgrad_ok = (not needs_input_grad) or _planes_ns(conv.out_channels, conv.in_channels, ks, False) == ns
    wgrad_ok = _two(ns) and _wgrad_planes_ok(B, conv.in_channels, H, W, conv.out_channels, ks)
    return ns, not (dgrad_ok and wgrad_ok and conv.bias is None)


def _planes_ns(Ci, Co, KS, up2):
    """Number of bf16 planes when a forward-type conv GEMM (Ci -> Co) runs on pre-split operands, else 0."""
    ns = _NS[_CONV_MATH[0]]
    if not ns or not _PLANES[0]:
        return 0
    if not up2 and _two(ns) and _SMALL_PLANES[0] and lib.itcv_conv2d_small_cout_bf16p_supported(Ci, Co, KS):
        return ns    # the 5x5 predict conv / stem data-gradient on the matrix cores (two-plane formats)
    if not up2 and (lib.itcv_conv2d_small_cout_supported(Co, KS) or lib.itcv_conv2d_small_cin_supported(Ci, KS)):
        return 0
    return ns if lib.itcv_conv2d_bf16s_supported(Ci, Co, KS) else 0


def _wgrad5_mode(Ci, H, W, Co, KS, up2):
    """'stem' / 'predict' when the 5x5 weight gradient with a 3-channel side runs on the matrix cores (bf16x3)."""
    if up2 or KS != 5 or not _two(_NS[_CONV_MATH[0]]) or not (_PLANES[0] and _SMALL_PLANES[0]):
        return None
    if Ci <= 3 and lib.itcv_conv2d_wgrad5_bf16p_supported(Ci, Co, H, W):
        return "stem"
    if Co <= 3 and lib.itcv_conv2d_wgrad5_bf16p_supported(Co, Ci, H, W):
        return "predict"
    return None


def conv_wgrad5_planes(small, big_planes, B, Cs, H, W, stem, out=None, accumulate=False, ns=2):
    """5x5 weight gradient with a <= 3-channel side from the planes of the 64-channel side.  fp16 form: the predict
    conv's small side is a gradient tensor and gets a device-side scale; the stem's is the input image (scale 1)."""
    dw = out if out is not None else torch.empty((64, Cs, 5, 5) if stem else (Cs, 64, 5, 5), dtype=F32, device=small.device)
    nws = lib.itcv_conv2d_wgrad5_bf16p_workspace(B, H)
    ws = _ws(nws, small.device)

    amax = absmax_parts(small) if (ns == F16X2 and not stem) else None
    call("itcv_conv2d_wgrad5_bf16p", ptr(small), ptr(big_planes), ptr(dw), B, Cs, H, W, int(stem), int(ns), ptr(amax),
         int(accumulate), ptr(ws), nws, stream())
    return dw


def _wgrad_planes_ok(B, Ci, H, W, Co, KS):
    return bool(_two(_NS[_CONV_MATH[0]]) and _PLANES[0] and lib.itcv_conv2d_wgrad_bf16p_supported(B, Ci, H, W, Co, KS))


class Conv2dFn(Function):
    """nn.Conv2d(stride 1, padding KS//2) forward/backward; ``up2`` reads the input through a
    virtual nearest x2 upsampling (models.py:284-286 fused into the consumer conv).

    In the split-bf16 modes the operands travel as pre-split planes: x is split once in forward (and
    kept for the weight gradient), dy once in backward (shared by data- and weight-gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, up2):
        x, weight = _f32c(x), _f32c(weight)
        B, Ci, Hs, Ws = x.shape
        Co, KS = weight.shape[0], weight.shape[2]
        H, W = (Hs * 2, Ws * 2) if up2 else (Hs, Ws)
        b = None if bias is None else _f32c(bias)
        ns = _planes_ns(Ci, Co, KS, up2)
        xp = None
        if ns:
            xp = planes_of(x, ns)
            y = conv_apply_planes(xp, weight, weight, 0, b, B, Ci, H, W, Co, KS, up2, ns, want_stats=True)
        else:
            y = conv_apply(_require_fp32(x, "Conv2dFn.forward"), weight, weight, 0, b, B, Ci, H, W, Co, KS, up2)
        wg_planes = _wgrad_planes_ok(B, Ci, H, W, Co, KS)
        keep_xp = xp if (_two(ns) and (wg_planes or _wgrad5_mode(Ci, H, W, Co, KS, up2) == "predict")) else None
        ctx.save_for_backward(None if keep_xp is not None else _require_fp32(x, "Conv2dFn.forward (saved input)"),
                              weight, bias, keep_xp)
        ctx.cfg = (B, Ci, H, W, Co, KS, up2, bias is not None, (Hs, Ws))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, bias, xp = ctx.saved_tensors
        B, Ci, H, W, Co, KS, up2, has_bias, (Hs, Ws) = ctx.cfg
        dy = _f32c(dy)
        dx = dw = db = None
        ns_d = _planes_ns(Co, Ci, KS, False) if ctx.needs_input_grad[0] else 0
        wg_planes = ctx.needs_input_grad[1] and _wgrad_planes_ok(B, Ci, H, W, Co, KS)
        dyp = None
        fmt2 = F16X2 if _NS[_CONV_MATH[0]] == F16X2 else 2       # the two-plane format of the current mode
        if ns_d or wg_planes:
            dyp = planes_of(dy, ns_d if ns_d else fmt2, gradient=True)
        if ctx.needs_input_grad[0]:
            if ns_d:
                dx = conv_apply_planes(dyp, weight, weight, 1, None, B, Co, H, W, Ci, KS, False, ns_d)
            else:
                dx = conv_apply(_require_fp32(dy, "Conv2dFn.backward"), weight, weight, 1, None, B, Co, H, W, Ci, KS, False)
            if up2:
                lo = torch.empty((B, Ci, H // 2, W // 2), dtype=F32, device=dy.device)
                call("itcv_upsample2_bwd", ptr(dx), ptr(lo), B * Ci, H // 2, W // 2, stream())
                dx = lo
        wg5 = _wgrad5_mode(Ci, H, W, Co, KS, up2) if ctx.needs_input_grad[1] else None
        if wg5 == "predict" and xp is None and x is None:
            wg5 = None
        if wg5 is not None:
            tgt = _grad_target(weight)
            if wg5 == "stem":
                small, big = _require_fp32(x, "Conv2dFn.backward (5x5 weight gradient)"), (dyp if dyp is not None else planes_of(dy, fmt2, gradient=True))
            else:
                small, big = _require_fp32(dy, "Conv2dFn.backward (5x5 weight gradient)"), (xp if xp is not None else split_planes(x, fmt2))
            cs = Ci if wg5 == "stem" else Co
            if tgt is not None:
                conv_wgrad5_planes(small, big, B, cs, H, W, wg5 == "stem", out=tgt, accumulate=True, ns=fmt2)
            else:
                dw = conv_wgrad5_planes(small, big, B, cs, H, W, wg5 == "stem", ns=fmt2)
        elif ctx.needs_input_grad[1]:
            tgt = _grad_target(weight)
            if wg_planes and (ns_d in (0, fmt2)):
                if xp is None:
                    xp = split_planes(x, fmt2)
                if tgt is not None:
                    conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, KS, up2, out=tgt, accumulate=True, ns=fmt2)
                else:
                    dw = conv_wgrad_planes(xp, dyp, B, Ci, H, W, Co, KS, up2, ns=fmt2)
            else:
                if x is None:
                    raise abi.HipExtensionError("Conv2dFn.backward: conv math mode changed between forward and backward")
                _require_fp32(dy, "Conv2dFn.backward (weight gradient)")
                if tgt is not None:
                    conv_wgrad_raw(x, dy, B, Ci, H, W, Co, KS, up2, out=tgt, accumulate=True)
                else:
                    dw = conv_wgrad_raw(x, dy, B, Ci, H, W, Co, KS, up2)
        if has_bias and ctx.needs_input_grad[2]:
            db = bias_grad_raw(_require_fp32(dy, "Conv2dFn.backward (bias gradient)"), B, Co, H * W, _grad_target(bias))
        return dx, dw, db, None


class LinearFn(Function):
    """nn.Linear on the skinny fp32 MFMA GEMMs (itcv_linear_*): exact fp32 in every conv-math mode."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight = _f32c(x), _f32c(weight)
        B, K = x.shape
        N = weight.shape[0]
        y = torch.empty((B, N), dtype=F32, device=x.device)
        nws = lib.itcv_linear_workspace(B, K, N)
        ws = _ws(nws, x.device) if nws else None
        call("itcv_linear_fwd", ptr(x), ptr(weight), ptr(None if bias is None else _f32c(bias)), ptr(y), B, K, N, ptr(ws),
             nws, stream())
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (B, K, N, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        B, K, N, has_bias = ctx.cfg
        dy = _f32c(dy)
        dx = dw = db = None
        nws = lib.itcv_linear_workspace(B, K, N)
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, K), dtype=F32, device=dy.device)
            ws = _ws(nws, dy.device) if nws else None
            call("itcv_linear_dgrad", ptr(dy), ptr(weight), ptr(dx), B, K, N, ptr(ws), nws, stream())
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(weight)
            dw = None if tgt is not None else torch.empty((N, K), dtype=F32, device=dy.device)
            ws = _ws(nws, dy.device) if nws else None
            call("itcv_linear_wgrad", ptr(dy), ptr(x), ptr(tgt if tgt is not None else dw), B, K, N,
                 int(tgt is not None), ptr(ws), nws, stream())
        if has_bias and ctx.needs_input_grad[2]:
            db = bias_grad_raw(dy, B, N, 1, _grad_target(bias))
        return dx, dw, db


# ------------------------------------------------------------------ BatchNorm + LeakyReLU (+pool)
def _world(group):
    return dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1


class BnActFn(Function):
    """y = pool(lrelu(BN_train(x) (+skip), slope)).  slope=1 -> plain BatchNorm; pool in {0,1}.
    ``group``: process group for Sync-BN (moments all-reduced in fp64).
    ``bn_groups`` = G > 1: the batch holds G independent network passes of B/G images each (the solvers push passes
    that share their weights through the conv GEMMs as ONE batch); every pass is normalised with its own batch
    statistics and advances the running buffers on its own, in order -- exactly G separate BatchNorm calls
    (models.py:37-38), issued here as G launches on sub-batches of the same tensors."""

    @staticmethod
    def forward(ctx, x, gamma, beta, skip, running_mean, running_var, nbt, eps, momentum, slope, pool, training,
                group, out_planes=0, grad_planes=0, out_fp32=True, grad_fp32=True, bn_groups=1):
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        skip = None if skip is None else _f32c(skip)
        B, C, H, W = x.shape
        G = int(bn_groups) if training else 1
        if G < 1 or B % G:
            raise abi.HipExtensionError(f"BatchNorm groups: batch {B} is not {G} equal passes")
        Bg = B // G
        dev = x.device
        mean = torch.empty((G, C), dtype=F32, device=dev)
        rstd = torch.empty((G, C), dtype=F32, device=dev)
        world = _world(group) if training else 1
        oshape = (B, C, H // 2, W // 2) if pool else (B, C, H, W)
        y = torch.empty(oshape, dtype=F32, device=dev)
        yp, pstride = None, 0
        if F16X2 in (out_planes, grad_planes) and not (training and world == 1):
            # fp16 planes carry one scale record per tensor, written by the single-call training forms (itcv_bn_train_fwd /
            # _bwd); the per-group eval / Sync-BN calls below hand the consumers fp32 tensors instead (split on demand)
            out_planes = 0 if out_planes == F16X2 else out_planes
            grad_planes = 0 if grad_planes == F16X2 else grad_planes
            out_fp32 = grad_fp32 = True
        if out_planes and lib.itcv_bn_act_planes_supported(C, H, W, int(pool)):
            yp = torch.empty(lib.itcv_planes_bytes(B, C, oshape[2] * oshape[3], out_planes) // 4, dtype=torch.int32,
                             device=dev)
            pstride = B * (C // 8) * oshape[2] * oshape[3]            # chunks between planes of the WHOLE tensor
        write_y = out_fp32 or yp is None or _POISON[0]
        nws = lib.itcv_bn_workspace(Bg, C, H * W) * G      # every group's partial sums (one launch for all groups)
        ts = _tile_stats_of(x, B, G, H * W) if (training and world == 1) else None
        if training and world == 1:
            # statistics + apply for all G groups in one call (small layers: one statistics launch walking the groups in
            # order + one apply launch; large layers: the groups one after the other inside the library)
            ws = _ws(nws, dev)
            call("itcv_bn_train_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(skip), ptr(y) if write_y else None, ptr(yp),
                 int(out_planes), Bg, C, H, W, float(slope), int(pool), float(eps), float(momentum), ptr(running_mean),
                 ptr(running_var), ptr(nbt), ptr(mean), ptr(rstd), ptr(ws), nws, pstride,
                 None if ts is None else ts[0].data_ptr(), ts[1] if ts else 0, ts[2] if ts else 0, G, stream())
        for g in (range(G) if not (training and world == 1) else ()):
            r = slice(g * Bg, (g + 1) * Bg)
            xg, yg = x[r], (y[r] if write_y else None)
            sg = None if skip is None else skip[r]
            ypg = None if yp is None else yp[g * Bg * (C // 8) * oshape[2] * oshape[3] * 4:]    # int32 elements
            if training:
                ws = _ws(nws, dev)
                sums = torch.empty((2 * C,), dtype=torch.float64, device=dev)
                call("itcv_bn_moments", ptr(xg), ptr(sums), Bg, C, H * W, ptr(ws), nws, stream())
                dist.all_reduce(sums, group=group)
                call("itcv_bn_finalize", ptr(sums), float(Bg * H * W * world), float(eps), float(momentum),
                     ptr(running_mean), ptr(running_var), ptr(nbt), ptr(mean[g]), ptr(rstd[g]), C, stream())
            else:
                call("itcv_bn_eval_stats", ptr(running_mean), ptr(running_var), float(eps), ptr(mean[g]), ptr(rstd[g]), C,
                     stream())
            call("itcv_bn_act_fwd", ptr(xg), ptr(mean[g]), ptr(rstd[g]), ptr(gamma), ptr(beta), ptr(sg), ptr(yg), Bg, C, H, W,
                 float(slope), int(pool), ptr(ypg), int(out_planes), pstride, stream())
        if yp is not None:
            if _POISON[0] and not out_fp32:
                y.fill_(float("nan"))
            _tag_planes(y, yp, out_planes, bool(out_fp32))
        ctx.save_for_backward(x, gamma, beta, mean, rstd, skip)
        ctx.params = (gamma, beta)
        if not lib.itcv_bn_act_planes_supported(C, H, W, 0):
            grad_planes = 0
        ctx.cfg = (B, C, H, W, float(slope), int(pool), bool(training), group, world, int(grad_planes),
                   bool(grad_fp32) or not grad_planes, G)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, nbt) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd, skip = ctx.saved_tensors
        B, C, H, W, slope, pool, training, group, world, grad_planes, grad_fp32, G = ctx.cfg
        if not training:
            raise abi.HipExtensionError("BatchNorm backward in eval mode is not part of the training hot path")
        dy = _f32c(dy)
        dev = x.device
        Bg = B // G
        nws = lib.itcv_bn_workspace(Bg, C, H * W) * G
        # parameter gradients come out of the reduce launch: either added straight into .grad
        # (solver mode) or into fresh tensors handed to autograd
        tg = _grad_target(gamma) if ctx.needs_input_grad[1] else None
        tb = _grad_target(beta) if ctx.needs_input_grad[2] else None
        direct = tg is not None and tb is not None
        dgamma = dbeta = None
        if direct:
            pg, pb = tg, tb
        else:
            dgamma = torch.empty_like(gamma) if ctx.needs_input_grad[1] else None
            dbeta = torch.empty_like(beta) if ctx.needs_input_grad[2] else None
            pg, pb = dgamma, dbeta
        dx = torch.empty_like(x)
        dskip = torch.empty_like(x) if (skip is not None and ctx.needs_input_grad[3]) else None
        dxp, pstride = None, 0
        if grad_planes:
            dxp = torch.empty(lib.itcv_planes_bytes(B, C, H * W, grad_planes) // 4, dtype=torch.int32, device=dev)
            pstride = B * (C // 8) * H * W
        write_dx = grad_fp32 or dxp is None or _POISON[0]
        if world == 1:
            ws = _ws(nws, dev)
            local = torch.empty((G, 2 * C), dtype=torch.float64, device=dev)
            call("itcv_bn_train_bwd", ptr(x), ptr(dy), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(skip), ptr(local),
                 ptr(dx) if write_dx else None, ptr(dskip), ptr(dxp), grad_planes, ptr(pg), ptr(pb), 1 if direct else 0, Bg,
                 C, H, W, slope, pool, 0, ptr(ws), nws, pstride, G, stream())
        for g in (range(G) if world != 1 else ()):
            r = slice(g * Bg, (g + 1) * Bg)
            acc = 1 if (direct or g > 0) else 0       # the groups' parameter gradients add up
            xg, dyg = x[r], dy[r]
            sg = None if skip is None else skip[r]
            dxg = dx[r] if write_dx else None
            dsg = None if dskip is None else dskip[r]
            dxpg = None if dxp is None else dxp[g * Bg * (C // 8) * H * W * 4:]
            ws = _ws(nws, dev)
            local = torch.empty((2 * C,), dtype=torch.float64, device=dev)
            call("itcv_bn_act_bwd_reduce", ptr(xg), ptr(dyg), ptr(mean[g]), ptr(rstd[g]), ptr(gamma), ptr(beta), ptr(sg),
                 ptr(local), ptr(pg), ptr(pb), acc, Bg, C, H, W, slope, pool, 0, ptr(ws), nws, stream())
            total = local.clone()
            dist.all_reduce(total, group=group)
            call("itcv_bn_act_bwd_apply", ptr(xg), ptr(dyg), ptr(mean[g]), ptr(rstd[g]), ptr(gamma), ptr(beta), ptr(sg),
                 ptr(total), None, float(Bg * H * W * world), ptr(dxg), ptr(dsg), None, None, 0, Bg, C, H, W, slope,
                 pool, 0, ptr(dxpg), grad_planes, pstride, stream())
        if dxp is not None:
            if _POISON[0] and not grad_fp32:
                dx.fill_(float("nan"))
            _tag_planes(dx, dxp, grad_planes, bool(grad_fp32))
        return (dx, dgamma, dbeta, dskip) + (None,) * 14


# ------------------------------------------------------------------ pointwise / resampling
class LeakyReluFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = _f32c(x)
        y = torch.empty_like(x)
        call("itcv_lrelu_fwd", ptr(x), ptr(y), x.numel(), float(slope), stream())
        ctx.save_for_backward(x)
        ctx.slope = float(slope)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        call("itcv_lrelu_bwd", ptr(x), ptr(dy), ptr(dx), x.numel(), ctx.slope, stream())
        return dx, None


class SigmoidFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        call("itcv_sigmoid_fwd", ptr(x), ptr(y), x.numel(), stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(y)
        call("itcv_sigmoid_bwd", ptr(y), ptr(dy), ptr(dx), y.numel(), stream())
        return dx


class AvgPool2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=F32, device=x.device)
        call("itcv_avgpool2_fwd", ptr(x), ptr(y), B * C, H, W, stream())
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _f32c(dy)
        dx = torch.empty((B, C, H, W), dtype=F32, device=dy.device)
        call("itcv_avgpool2_bwd", ptr(dy), ptr(dx), B * C, H, W, stream())
        return dx


class Upsample2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), dtype=F32, device=x.device)
        call("itcv_upsample2_fwd", ptr(x), ptr(y), B * C, H, W, stream())
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = _f32c(dy)
        dx = torch.empty((B, C, H, W), dtype=F32, device=dy.device)
        call("itcv_upsample2_bwd", ptr(dy), ptr(dx), B * C, H, W, stream())
        return dx


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        out = torch.empty_like(a)
        call("itcv_add", ptr(a), ptr(b), ptr(out), a.numel(), stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


# ------------------------------------------------------------------ latent math
class ReparamFn(Function):
    """ops.py:166-185 with the N(0,1) draw as an explicit input."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _f32c(mu), _f32c(logvar), _f32c(eps)
        z = torch.empty_like(mu)
        call("itcv_reparam_fwd", ptr(mu), ptr(logvar), ptr(eps), ptr(z), mu.numel(), stream())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = _f32c(dz)
        dmu, dlv = torch.empty_like(dz), torch.empty_like(dz)
        call("itcv_reparam_bwd", ptr(dz), ptr(logvar), ptr(eps), ptr(dmu), ptr(dlv), dz.numel(), stream())
        return dmu, dlv, None


class KlRowsFn(Function):
    """ops.py:161-163 -> [B]."""

    @staticmethod
    def forward(ctx, logvar, mu):
        logvar, mu = _f32c(logvar), _f32c(mu)
        B, D = logvar.shape
        kl = torch.empty((B,), dtype=F32, device=mu.device)
        call("itcv_kl_rows_fwd", ptr(logvar), ptr(mu), ptr(kl), B, D, stream())
        ctx.save_for_backward(logvar, mu)
        return kl

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logvar, mu = ctx.saved_tensors
        B, D = logvar.shape
        g = _f32c(g)
        dlv, dmu = torch.empty_like(logvar), torch.empty_like(mu)
        call("itcv_kl_rows_bwd", ptr(g), ptr(logvar), ptr(mu), ptr(dlv), ptr(dmu), B, D, stream())
        return dlv, dmu


_REDUCTION = {"none": 0, "sum": 1, "mean": 2}


class KlLossFn(Function):
    """scale * ops.kl_divergence(logvar, mu, reduce) (ops.py:136-163 + the hook's beta, solvers/vae.py:63-77): one launch."""

    @staticmethod
    def forward(ctx, logvar, mu, reduction, scale):
        logvar, mu = _f32c(logvar), _f32c(mu)
        B, D = logvar.shape
        out = torch.empty((B,) if reduction == 0 else (), dtype=F32, device=mu.device)
        call("itcv_kl_loss_fwd", ptr(logvar), ptr(mu), ptr(out), B, D, reduction, float(scale), stream())
        ctx.save_for_backward(logvar, mu)
        ctx.cfg = (B, D, reduction, float(scale))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logvar, mu = ctx.saved_tensors
        B, D, reduction, scale = ctx.cfg
        dlv, dmu = torch.empty_like(logvar), torch.empty_like(mu)
        call("itcv_kl_loss_bwd", ptr(_f32c(g)), ptr(logvar), ptr(mu), ptr(dlv), ptr(dmu), B, D, reduction, scale, stream())
        return dlv, dmu, None, None


class TcKlFn(Function):
    """coef_tc * total_correlation + coef_kl * kl_divergence with the hook's reduction (solvers/tc.py:69-89), fused into
    the estimator's launches (forward: partials, finish [, reduce]; backward: rows, columns)."""

    @staticmethod
    def forward(ctx, z, mu_all, logvar, dataset_size, row_offset, coef_tc, coef_kl, reduction):
        z, mu_all, logvar = _f32c(z), _f32c(mu_all), _f32c(logvar)
        Bl, D = z.shape
        Bt = mu_all.shape[0]
        dev = z.device
        out = torch.empty((Bl,) if reduction == 0 else (), dtype=F32, device=dev)
        rows = torch.empty((Bl,), dtype=F32, device=dev) if reduction else None
        prodm = torch.empty((Bl,), dtype=F32, device=dev)
        logqz = torch.empty((Bl,), dtype=F32, device=dev)
        lse = torch.empty((Bl, D), dtype=F32, device=dev)
        sjoint = torch.empty((Bl, Bt), dtype=F32, device=dev)
        nws = lib.itcv_tc_fwd_workspace(Bl, Bt, D)
        ws = _ws(nws, dev)
        call("itcv_tc_kl_fwd", ptr(z), ptr(mu_all), ptr(logvar), ptr(out), ptr(rows), ptr(prodm), ptr(logqz), ptr(lse),
             ptr(sjoint), Bl, Bt, int(row_offset), D, int(dataset_size), float(coef_tc), float(coef_kl), reduction, ptr(ws),
             nws, stream())
        ctx.save_for_backward(z, mu_all, logvar, logqz, lse, sjoint)
        ctx.cfg = (int(dataset_size), int(row_offset), float(coef_tc), float(coef_kl), reduction)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        z, mu_all, logvar, logqz, lse, sjoint = ctx.saved_tensors
        n, off, ctc, ckl, reduction = ctx.cfg
        Bl, D = z.shape
        Bt = mu_all.shape[0]
        dz, dlv, dmu = torch.empty_like(z), torch.empty_like(logvar), torch.empty_like(mu_all)
        nws = lib.itcv_tc_bwd_workspace(Bl, Bt)
        ws = _ws(nws, z.device)
        call("itcv_tc_kl_bwd", ptr(_f32c(g)), ptr(z), ptr(mu_all), ptr(logvar), ptr(logqz), ptr(lse), ptr(sjoint), ptr(dz),
             ptr(dmu), ptr(dlv), Bl, Bt, off, D, n, ctc, ckl, reduction, ptr(ws), nws, stream())
        return dz, dmu, dlv, None, None, None, None, None


class ReconLossFn(Function):
    """scale * ops.reconstruction_loss(x, recon, loss_type, reduction) (ops.py:188-236 + the hook's beta): two launches
    forward, one backward; x is a constant."""

    @staticmethod
    def forward(ctx, x, recon, loss_type, reduction, scale):
        x, recon = _f32c(x), _f32c(recon)
        B = recon.shape[0]
        P = recon.numel() // B
        out = torch.empty((B,) if reduction == 0 else (), dtype=F32, device=recon.device)
        nws = lib.itcv_recon_workspace(B, P)
        ws = _ws(nws, recon.device)
        call("itcv_recon_loss_fwd", ptr(x), ptr(recon), ptr(out), B, P, loss_type, reduction, float(scale), ptr(ws), nws,
             stream())
        ctx.save_for_backward(x, recon)
        ctx.cfg = (B, P, loss_type, reduction, float(scale))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, recon = ctx.saved_tensors
        B, P, lt, reduction, scale = ctx.cfg
        d = torch.empty_like(recon)
        call("itcv_recon_loss_bwd", ptr(x), ptr(recon), ptr(_f32c(g)), ptr(d), B, P, lt, reduction, scale, stream())
        return None, d, None, None, None


class ExpElboFn(Function):
    """mean_j exp(c * (a_j + b_j)) (solvers/intro.py:102-103 with c = -2 * scale): one launch each way."""

    @staticmethod
    def forward(ctx, a, b, c):
        a, b = _f32c(a), _f32c(b)
        B = a.shape[0]
        out = torch.empty((), dtype=F32, device=a.device)
        w = torch.empty((B,), dtype=F32, device=a.device)
        call("itcv_exp_elbo_fwd", ptr(a), ptr(b), ptr(out), ptr(w), B, float(c), stream())
        ctx.save_for_backward(w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        da, db = torch.empty_like(w), torch.empty_like(w)
        call("itcv_exp_elbo_bwd", ptr(_f32c(g)), ptr(w), ptr(da), ptr(db), w.shape[0], stream())
        return da, db, None


class LinCombFn(Function):
    """sum_k weights[k] * terms[k] over up to 8 device scalars: the scalar arithmetic of a solver's loss in one launch."""

    @staticmethod
    def forward(ctx, weights, *terms):
        import ctypes
        n = len(terms)
        terms = [_f32c(t) for t in terms]
        if any(t.numel() != 1 for t in terms):
            raise abi.HipExtensionError("LinCombFn: every term must be a scalar")
        out = torch.empty((), dtype=F32, device=terms[0].device)
        tp = (ctypes.c_void_p * n)(*[ptr(t) for t in terms])
        wp = (ctypes.c_float * n)(*[float(w) for w in weights])
        call("itcv_lincomb_fwd", tp, wp, n, ptr(out), stream())
        ctx.weights = [float(w) for w in weights]
        ctx.shapes = [t.shape for t in terms]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        import ctypes
        n = len(ctx.weights)
        grads = torch.empty((n,), dtype=F32, device=g.device)
        wp = (ctypes.c_float * n)(*ctx.weights)
        call("itcv_lincomb_bwd", ptr(_f32c(g)), wp, n, ptr(grads), stream())
        return (None,) + tuple(grads[k].reshape(ctx.shapes[k]) for k in range(n))


def tc_components(z, mu_all, logvar, dataset_size, row_offset=0, flags=abi.TC_LIVE, with_joint=False):
    """(prodm[Bl], logqz[Bl], lse[Bl,D]) of the fused pairwise-density / sampling kernel (+ the joint terms
    S[Bl,Bt] the backward reads, with ``with_joint``)."""
    z, mu_all, logvar = _f32c(z), _f32c(mu_all), _f32c(logvar)
    Bl, D = z.shape
    Bt = mu_all.shape[0]
    dev = z.device
    prodm = torch.empty((Bl,), dtype=F32, device=dev)
    logqz = torch.empty((Bl,), dtype=F32, device=dev)
    lse = torch.empty((Bl, D), dtype=F32, device=dev)
    sjoint = torch.empty((Bl, Bt), dtype=F32, device=dev)
    nws = lib.itcv_tc_fwd_workspace(Bl, Bt, D)
    ws = _ws(nws, dev)
    call("itcv_tc_fwd", ptr(z), ptr(mu_all), ptr(logvar), ptr(prodm), ptr(logqz), ptr(lse), ptr(sjoint), Bl, Bt,
         int(row_offset), D, int(dataset_size), int(flags), ptr(ws), nws, stream())
    return (prodm, logqz, lse, sjoint) if with_joint else (prodm, logqz, lse)


class TcRowsFn(Function):
    """ops.py:52-89 (live estimator) per local sample: tc[j] = log q(z_j) - sum_l log q(z_jl).
    ``mu_all`` holds the means of the whole (global) batch; rows are the caller's samples."""

    @staticmethod
    def forward(ctx, z, mu_all, logvar, dataset_size, row_offset):
        prodm, logqz, lse, sjoint = tc_components(z, mu_all, logvar, dataset_size, row_offset, abi.TC_LIVE, True)
        ctx.save_for_backward(_f32c(z), _f32c(mu_all), _f32c(logvar), logqz, lse, sjoint)
        ctx.cfg = (int(dataset_size), int(row_offset))
        return logqz - prodm

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        z, mu_all, logvar, logqz, lse, sjoint = ctx.saved_tensors
        n, off = ctx.cfg
        Bl, D = z.shape
        Bt = mu_all.shape[0]
        g = _f32c(g)
        dz, dlv, dmu = torch.empty_like(z), torch.empty_like(logvar), torch.empty_like(mu_all)
        nws = lib.itcv_tc_bwd_workspace(Bl, Bt)
        ws = _ws(nws, z.device)
        call("itcv_tc_bwd", ptr(g), ptr(z), ptr(mu_all), ptr(logvar), ptr(logqz), ptr(lse), ptr(sjoint), ptr(dz), ptr(dmu),
             ptr(dlv), Bl, Bt, off, D, n, abi.TC_LIVE, ptr(ws), nws, stream())
        return dz, dmu, dlv, None, None


def _bc3(x, mu, logvar):
    """The three operands as (same-storage) views broadcast to one shape of at most three dimensions, with their element
    strides (0 on broadcast dimensions) as ctypes arrays."""
    import ctypes
    x, mu, logvar = (t if t.dtype == F32 else t.float() for t in (x, mu, logvar))
    xb, mb, lb = torch.broadcast_tensors(x, mu, logvar)
    shape = tuple(xb.shape)
    if len(shape) > 3:    # collapse the leading dimensions (needs dense operands there)
        xb, mb, lb = (t.contiguous().reshape(-1, *shape[-2:]) for t in (xb, mb, lb))
    while xb.dim() < 3:
        xb, mb, lb = xb.unsqueeze(0), mb.unsqueeze(0), lb.unsqueeze(0)
    arr = lambda v: (ctypes.c_int64 * 3)(*v)
    return shape, (xb, mb, lb), arr(xb.shape), arr(xb.stride()), arr(mb.stride()), arr(lb.stride())


class GaussLogDensityFn(Function):
    """ops.py:15-21 (``eps_density``) / ops.py:24-29 on broadcastable operands, materialised."""

    @staticmethod
    def forward(ctx, x, mu, logvar, eps_density):
        shape, views, dims, sx, sm, sl = _bc3(x, mu, logvar)
        out = torch.empty(tuple(views[0].shape), dtype=F32, device=views[0].device)
        for v in views:
            if not v.is_cuda:
                raise abi.HipExtensionError("HIP kernels need device tensors (got a CPU tensor); there is no CPU path")
        call("itcv_gauss_logdensity_fwd", views[0].data_ptr(), views[1].data_ptr(), views[2].data_ptr(), ptr(out), dims, sx,
             sm, sl, int(eps_density), stream())
        ctx.save_for_backward(x, mu, logvar)
        ctx.eps_density = int(eps_density)
        return out.reshape(shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, mu, logvar = ctx.saved_tensors
        shape, views, dims, sx, sm, sl = _bc3(x, mu, logvar)
        g = _f32c(g).reshape(tuple(views[0].shape))
        dx, dlv = torch.empty_like(g), torch.empty_like(g)
        call("itcv_gauss_logdensity_bwd", ptr(g), views[0].data_ptr(), views[1].data_ptr(), views[2].data_ptr(), ptr(dx),
             ptr(dlv), dims, sx, sm, sl, ctx.eps_density, stream())
        dx, dlv = dx.reshape(shape), dlv.reshape(shape)
        # un-broadcast (torch reduction: plumbing of this off-path helper, not step arithmetic)
        return (dx.sum_to_size(x.shape) if ctx.needs_input_grad[0] else None,
                (-dx).sum_to_size(mu.shape) if ctx.needs_input_grad[1] else None,
                dlv.sum_to_size(logvar.shape) if ctx.needs_input_grad[2] else None, None)


class SamplingFn(Function):
    """ops.py:104-115 / ops.py:92-101 on a materialised [B,B,D] log-density tensor -> (prodm[B], logqz[B])."""

    @staticmethod
    def forward(ctx, lp, dataset_size, weighted):
        lp = _f32c(lp)
        B, B2, D = lp.shape
        if B != B2:
            raise abi.HipExtensionError(f"sampling: log_qz_prob must be [B,B,D] (got {tuple(lp.shape)})")
        dev = lp.device
        prodm, logqz = torch.empty((B,), dtype=F32, device=dev), torch.empty((B,), dtype=F32, device=dev)
        lse, sj = torch.empty((B, D), dtype=F32, device=dev), torch.empty((B, B), dtype=F32, device=dev)
        call("itcv_sampling_fwd", ptr(lp), ptr(prodm), ptr(logqz), ptr(lse), ptr(sj), B, D, int(dataset_size), int(weighted),
             stream())
        ctx.save_for_backward(lp, lse, sj, logqz)
        ctx.cfg = (B, D, int(dataset_size), int(weighted))
        return prodm, logqz

    @staticmethod
    @once_differentiable
    def backward(ctx, gp, gq):
        lp, lse, sj, logqz = ctx.saved_tensors
        B, D, n, weighted = ctx.cfg
        dlp = torch.empty_like(lp)
        call("itcv_sampling_bwd", ptr(_f32c(gp)), ptr(_f32c(gq)), ptr(lp), ptr(lse), ptr(sj), ptr(logqz), ptr(dlp), B, D, n,
             weighted, stream())
        return dlp, None, None


def on_off_diag(x):
    x = _f32c(x)
    if x.dim() != 2 or not (x.shape[0] == x.shape[1] or x.shape[0] == 1):
        raise abi.HipExtensionError(f"on_off_diag: a [n,n] or [1,n] tensor is expected (got {tuple(x.shape)})")
    m, n = x.shape
    diag = torch.empty((min(m, n),), dtype=F32, device=x.device)
    off = torch.empty((m, n, n), dtype=F32, device=x.device)
    call("itcv_on_off_diag", ptr(x), ptr(diag), ptr(off), m, n, stream())
    return diag, off


def diag_logdensity_rows(z, mu, logvar):
    z, mu, logvar = _f32c(z), _f32c(mu), _f32c(logvar)
    B, D = z.shape
    a = torch.empty((B,), dtype=F32, device=z.device)
    b = torch.empty((B,), dtype=F32, device=z.device)
    call("itcv_diag_logdensity_rows", ptr(z), ptr(mu), ptr(logvar), ptr(a), ptr(b), B, D, stream())
    return a, b


class ReconRowsFn(Function):
    """ops.py:219-230: per-sample summed reconstruction error; x is treated as a constant."""

    @staticmethod
    def forward(ctx, x, recon, loss_type):
        x, recon = _f32c(x), _f32c(recon)
        B = recon.shape[0]
        P = recon.numel() // B
        rows = torch.empty((B,), dtype=F32, device=recon.device)
        nws = lib.itcv_recon_workspace(B, P)
        ws = _ws(nws, recon.device)
        call("itcv_recon_rows_fwd", ptr(x), ptr(recon), ptr(rows), B, P, loss_type, ptr(ws), nws, stream())
        ctx.save_for_backward(x, recon)
        ctx.cfg = (B, P, loss_type)
        return rows

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, recon = ctx.saved_tensors
        B, P, lt = ctx.cfg
        g = _f32c(g)
        d = torch.empty_like(recon)
        call("itcv_recon_rows_bwd", ptr(x), ptr(recon), ptr(g), ptr(d), B, P, lt, stream())
        return None, d, None
