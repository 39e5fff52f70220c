"""ctypes binding of libitcv_hip.so (the C ABI declared in include/itcv_hip.h).

This is the only place the Python host touches native code.  There is NO fallback: if the
shared library is missing or a symbol does not resolve, importing this module raises, and
every op in the package raises with it -- a CPU/eager substitute would silently void the
parity and performance claims.

PyTorch is used only as plumbing here: device memory (``tensor.data_ptr()``) and the current
HIP stream (``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
# ITCV_LIB: load another build of the same library (the -DITCV_DIAG diagnostic build of `make diag`, tools/abl.sh)
LIB_PATH = os.environ.get("ITCV_LIB") or os.path.join(PKG_ROOT, "lib", "libitcv_hip.so")
CSRC = os.path.join(PKG_ROOT, "csrc")
ABI_VERSION = 3

p, i32, i64, sz, f32, f64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t,
                             ctypes.c_float, ctypes.c_double)

# name -> (restype, [argtypes]); mirrors include/itcv_hip.h one to one
SIGNATURES = {
    "itcv_abi_version": (i32, []),
    "itcv_last_error": (ctypes.c_char_p, []),
    "itcv_set_option": (i32, [ctypes.c_char_p, i32]),
    "itcv_get_option": (i32, [ctypes.c_char_p]),
    "itcv_profile_begin": (i32, []),
    "itcv_profile_end": (i32, []),
    "itcv_profile_get": (i32, [i32, p, p, p]),
    "itcv_profile_clear": (i32, []),
    "itcv_conv2d_packed_weight_elems": (sz, [i32, i32, i32, i32]),
    "itcv_conv2d_pack_weight": (i32, [p, p, i32, i32, i32, i32, p]),
    "itcv_conv2d_fwd_workspace": (sz, [i32] * 6),
    "itcv_conv2d_fwd": (i32, [p, p, p, p, i32, i32, i32, i32, i32, i32, i32, p, sz, p]),
    "itcv_conv2d_bf16s_supported": (i32, [i32, i32, i32]),
    "itcv_conv2d_packed_weight_bytes_bf16s": (sz, [i32] * 5),
    "itcv_conv2d_pack_weight_bf16s": (i32, [p, p, i32, i32, i32, i32, i32, p]),
    "itcv_pack_desc_bytes": (sz, []),
    "itcv_conv2d_pack_desc_bf16s": (i32, [p, p, p, i32, i32, i32, i32, i32, i32]),
    "itcv_conv2d_pack_weights_bf16s": (i32, [p, i32, i32, i32, p]),
    "itcv_conv2d_fwd_bf16s_workspace": (sz, [i32] * 6),
    "itcv_conv2d_fwd_bf16s": (i32, [p, p, p, p] + [i32] * 8 + [p, sz, p]),
    "itcv_planes_bytes": (sz, [i32] * 4),
    "itcv_split_planes": (i32, [p, p, i32, i32, i32, i32, p]),
    "itcv_absmax": (i32, [p, sz, p, p]),
    "itcv_split_planes_scaled": (i32, [p, p, i32, i32, i32, i32, p, p]),
    "itcv_conv2d_fwd_bf16p_workspace": (sz, [i32] * 7),
    "itcv_conv2d_fwd_bf16p": (i32, [p, p, p, p] + [i32] * 8 + [p, sz, p]),
    "itcv_conv2d_fwd_bf16p_stat_tiles": (i32, [i32] * 7),
    "itcv_conv2d_fwd_bf16p_st": (i32, [p, p, p, p] + [i32] * 8 + [p, p, sz, p]),
    "itcv_conv2d_wgrad_bf16p_supported": (i32, [i32] * 6),
    "itcv_conv2d_wgrad_bf16p_workspace": (sz, [i32] * 6),
    "itcv_conv2d_wgrad_bf16p": (i32, [p, p, p] + [i32] * 9 + [p, sz, p]),
    "itcv_conv2d_wgrad_bf16p_slabs": (i32, [i32] * 6),
    "itcv_wgrad_reduce_desc_bytes": (sz, []),
    "itcv_wgrad_reduce_desc": (i32, [p, p, p, i32, p, i32, i32, i32, i32]),
    "itcv_wgrad_reduce_many": (i32, [p, i32, i32, p]),
    "itcv_wgrad_reduce_max_descs": (i32, []),
    "itcv_linear_workspace": (sz, [i32] * 3),
    "itcv_linear_fwd": (i32, [p, p, p, p, i32, i32, i32, p, sz, p]),
    "itcv_linear_dgrad": (i32, [p, p, p, i32, i32, i32, p, sz, p]),
    "itcv_linear_wgrad": (i32, [p, p, p, i32, i32, i32, i32, p, sz, p]),
    "itcv_conv2d_wgrad5_bf16p_supported": (i32, [i32] * 4),
    "itcv_conv2d_wgrad5_bf16p_workspace": (sz, [i32, i32]),
    "itcv_conv2d_wgrad5_bf16p": (i32, [p, p, p] + [i32] * 6 + [p, i32, p, sz, p]),
    "itcv_conv2d_small_cout_supported": (i32, [i32, i32]),
    "itcv_conv2d_small_cout_bf16p_supported": (i32, [i32, i32, i32]),
    "itcv_conv2d_small_cout_fwd_bf16p": (i32, [p, p, p, p] + [i32] * 8 + [p]),
    "itcv_conv2d_small_cout_fwd": (i32, [p, p, p, p] + [i32] * 7 + [p]),
    "itcv_conv2d_small_cin_supported": (i32, [i32, i32]),
    "itcv_conv2d_small_cin_fwd": (i32, [p, p, p, p] + [i32] * 7 + [p]),
    "itcv_conv2d_small_cin_bf16x3_supported": (i32, [i32, i32, i32, i32]),
    "itcv_conv2d_small_cin_fwd_bf16x3": (i32, [p, p, p, p] + [i32] * 8 + [p, p]),
    "itcv_conv2d_wgrad_bf16s_supported": (i32, [i32] * 5),
    "itcv_conv2d_wgrad_bf16s": (i32, [p, p, p] + [i32] * 8 + [p, sz, p]),
    "itcv_conv2d_wgrad_workspace": (sz, [i32] * 6),
    "itcv_conv2d_wgrad": (i32, [p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p, sz, p]),
    "itcv_conv2d_fwd_variant": (i32, [i32] * 7),
    "itcv_conv2d_wgrad_variant": (i32, [i32] * 7),
    "itcv_bias_grad_workspace": (sz, [i32, i32, i32]),
    "itcv_bias_grad": (i32, [p, p, i32, i32, i32, i32, p, sz, p]),
    "itcv_bn_workspace": (sz, [i32, i32, i32]),
    "itcv_bn_moments": (i32, [p, p, i32, i32, i32, p, sz, p]),
    "itcv_bn_train_stats": (i32, [p, i32, i32, i32, f32, f32, p, p, p, p, p, p, sz, p]),
    "itcv_bn_finalize": (i32, [p, f64, f32, f32, p, p, p, p, p, i32, p]),
    "itcv_bn_eval_stats": (i32, [p, p, f32, p, p, i32, p]),
    "itcv_bn_act_planes_supported": (i32, [i32] * 4),
    "itcv_bn_act_fwd": (i32, [p, p, p, p, p, p, p, i32, i32, i32, i32, f32, i32, p, i32, sz, p]),
    "itcv_bn_act_bwd_reduce": (i32, [p, p, p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, f32, i32, i32, p, sz,
                                     p]),
    "itcv_bn_act_bwd_apply": (i32, [p, p, p, p, p, p, p, p, p, f64, p, p, p, p, i32, i32, i32, i32, i32, f32,
                                    i32, i32, p, i32, sz, p]),
    "itcv_bn_train_fwd": (i32, [p, p, p, p, p, p, i32, i32, i32, i32, i32, f32, i32, f32, f32, p, p, p, p, p, p, sz, sz, p, i32, i32,
                                i32, p]),
    "itcv_bn_train_bwd": (i32, [p, p, p, p, p, p, p, p, p, p, p, i32, p, p, i32, i32, i32, i32, i32, f32, i32, i32, p,
                                sz, sz, i32, p]),
    "itcv_lrelu_fwd": (i32, [p, p, sz, f32, p]),
    "itcv_lrelu_bwd": (i32, [p, p, p, sz, f32, p]),
    "itcv_sigmoid_fwd": (i32, [p, p, sz, p]),
    "itcv_sigmoid_bwd": (i32, [p, p, p, sz, p]),
    "itcv_avgpool2_fwd": (i32, [p, p, i32, i32, i32, p]),
    "itcv_avgpool2_bwd": (i32, [p, p, i32, i32, i32, p]),
    "itcv_upsample2_fwd": (i32, [p, p, i32, i32, i32, p]),
    "itcv_upsample2_bwd": (i32, [p, p, i32, i32, i32, p]),
    "itcv_add": (i32, [p, p, p, sz, p]),
    "itcv_reparam_fwd": (i32, [p, p, p, p, sz, p]),
    "itcv_reparam_bwd": (i32, [p, p, p, p, p, sz, p]),
    "itcv_kl_rows_fwd": (i32, [p, p, p, i32, i32, p]),
    "itcv_kl_rows_bwd": (i32, [p, p, p, p, p, i32, i32, p]),
    "itcv_tc_fwd_workspace": (sz, [i32, i32, i32]),
    "itcv_tc_fwd": (i32, [p, p, p, p, p, p, p, i32, i32, i32, i32, i64, i32, p, sz, p]),
    "itcv_tc_bwd_workspace": (sz, [i32, i32]),
    "itcv_tc_bwd": (i32, [p, p, p, p, p, p, p, p, p, p, i32, i32, i32, i32, i64, i32, p, sz, p]),
    "itcv_tc_kl_fwd": (i32, [p] * 9 + [i32, i32, i32, i32, i64, f32, f32, i32, p, sz, p]),
    "itcv_tc_kl_bwd": (i32, [p] * 10 + [i32, i32, i32, i32, i64, f32, f32, i32, p, sz, p]),
    "itcv_kl_loss_fwd": (i32, [p, p, p, i32, i32, i32, f32, p]),
    "itcv_kl_loss_bwd": (i32, [p, p, p, p, p, i32, i32, i32, f32, p]),
    "itcv_diag_logdensity_rows": (i32, [p, p, p, p, p, i32, i32, p]),
    "itcv_gauss_logdensity_fwd": (i32, [p, p, p, p, p, p, p, p, i32, p]),
    "itcv_gauss_logdensity_bwd": (i32, [p, p, p, p, p, p, p, p, p, p, i32, p]),
    "itcv_sampling_fwd": (i32, [p, p, p, p, p, i32, i32, i64, i32, p]),
    "itcv_sampling_bwd": (i32, [p, p, p, p, p, p, p, i32, i32, i64, i32, p]),
    "itcv_on_off_diag": (i32, [p, p, p, i32, i32, p]),
    "itcv_recon_workspace": (sz, [i32, sz]),
    "itcv_recon_rows_fwd": (i32, [p, p, p, i32, sz, i32, p, sz, p]),
    "itcv_recon_rows_bwd": (i32, [p, p, p, p, i32, sz, i32, p]),
    "itcv_recon_loss_fwd": (i32, [p, p, p, i32, sz, i32, i32, f32, p, sz, p]),
    "itcv_recon_loss_bwd": (i32, [p, p, p, p, i32, sz, i32, i32, f32, p]),
    "itcv_exp_elbo_fwd": (i32, [p, p, p, p, i32, f32, p]),
    "itcv_exp_elbo_bwd": (i32, [p, p, p, p, i32, p]),
    "itcv_lincomb_fwd": (i32, [p, p, i32, p, p]),
    "itcv_lincomb_bwd": (i32, [p, p, i32, p, p]),
    "itcv_sumsq_workspace": (sz, [sz]),
    "itcv_sumsq": (i32, [p, sz, p, p, sz, p]),
    "itcv_clip_coef": (i32, [p, i32, f64, p, p, p]),
    "itcv_scale_by_dev": (i32, [p, sz, p, p]),
    "itcv_adam_step": (i32, [p, p, p, p, sz, f32, f32, f32, f32, i32, p]),
    "itcv_adam_step_dev": (i32, [p, p, p, p, sz, f32, f32, f32, f32, p, p]),
    "itcv_fill": (i32, [p, sz, f32, p]),
    "itcv_hflip": (i32, [p, p, p, i32, i32, i32, p]),
}

TC_VAR_FROM_ROW, TC_EPS_DENSITY, TC_WEIGHTED = 1, 2, 4
TC_LIVE = TC_VAR_FROM_ROW | TC_EPS_DENSITY
LOSS_TYPES = {"mse": 0, "l1": 1, "bce": 2}


class HipExtensionError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libitcv_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout[-4000:], out.stderr[-4000:])
    if out.returncode:
        raise HipExtensionError("building libitcv_hip.so failed (see output above)")
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "There is no CPU fallback for the HIP hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.itcv_abi_version() != ABI_VERSION:
        raise HipExtensionError(f"ABI version mismatch: library {lib.itcv_abi_version()} != binding {ABI_VERSION}")
    return lib


class _Lib:
    """The loaded library with its pure query entry points (``*_supported``, ``*_workspace``, ``*_bytes``, ``*_elems``,
    ``*_variant``: integers in, integer out, no device work) memoised: an eager step asks several thousand such
    questions, and a dict hit is ~10x cheaper than a ctypes call."""

    _PURE = ("_supported", "_workspace", "_bytes", "_elems", "_variant", "_stat_tiles", "_slabs", "_max_descs")

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name.endswith(self._PURE):
            memo = {}

            def cached(*args, _fn=fn, _memo=memo):
                r = _memo.get(args)
                if r is None:
                    r = _memo[args] = _fn(*args)
                return r

            fn = cached
        object.__setattr__(self, name, fn)
        return fn


lib = _Lib(_load())


def last_error():
    return lib.itcv_last_error().decode()


def check(rc, exc=RuntimeError):
    if rc:
        raise exc(last_error())


def stream():
    """Raw handle of torch's current HIP stream (torch.cuda.current_stream() builds a Stream object: ~9 us a call)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def ptr(t):
    """Device pointer of a dense fp32/fp64/int64 CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipExtensionError("HIP kernels need device tensors (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise HipExtensionError("HIP kernels need contiguous tensors")
    return t.data_ptr()


def call(name, *args):
    check(getattr(lib, name)(*args))
