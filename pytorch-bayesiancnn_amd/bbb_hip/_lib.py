"""ctypes binding of libbbb_hip.so (C ABI in include/bbb_hip.h).

The shared object is built in-tree by ``build.sh`` / ``__graft_entry__.build()``.  There is NO fallback:
if the library is missing or a symbol does not resolve, importing the compute ops raises.  torch is
imported first on purpose: libbbb_hip.so needs libamdhip64.so.7 and must bind to the HIP runtime torch
already loaded, so that device pointers and streams are shared.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBB_HIP_LIB") or os.path.join(_HERE, "libbbb_hip.so")   # env override: experiments only

ABI_VERSION = 13
MAX_SEGMENTS = 16
SIGMA_SQUARED = 1
KL_TEXTBOOK = 2
GW_MEAN_ONLY = 4

c_void_p, c_int, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
c_u64, c_u32, c_i64, c_i32 = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64, ctypes.c_int32


class Segment(ctypes.Structure):
    _fields_ = [("mu", c_void_p), ("rho", c_void_p), ("w", c_void_p), ("sigma", c_void_p), ("eps", c_void_p),
                ("n", c_i64), ("draw_stride", c_i64), ("stream_id", c_u32), ("w_row_len", c_u32), ("w_taps", c_u32), ("w_tm_cin", c_u32)]


class AdamSegment(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_i64)]


class FlipSeg(ctypes.Structure):
    _fields_ = [("w0", c_void_p), ("w1", c_void_p), ("out", c_void_p), ("draws", c_i64), ("cout", c_i32), ("cin", c_i32),
                ("khkw", c_i32), ("reserved", c_i32)]


class ConvDesc(ctypes.Structure):
    _fields_ = [("batch", c_i32), ("cin", c_i32), ("h", c_i32), ("w", c_i32), ("cout", c_i32), ("kh", c_i32),
                ("kw", c_i32), ("stride_h", c_i32), ("stride_w", c_i32), ("pad_h", c_i32), ("pad_w", c_i32),
                ("dil_h", c_i32), ("dil_w", c_i32), ("draws", c_i32), ("x_draw_stride", c_i64),
                ("w_draw_stride", c_i64), ("b_draw_stride", c_i64), ("act", c_i32),
                ("unit_div", c_i32), ("unit_off", c_i32), ("x_unit_mod", c_i32), ("w_row_pitch", c_i32), ("b_offset", c_i32),
                ("x_unit_div", c_i32), ("x_unit_off", c_i32), ("pool", c_i32), ("w_tap_major", c_i32)]


_SIGNATURES = {
    "bbb_reparam_kl_fwd": (c_int, [ctypes.POINTER(Segment), c_int, c_int, c_float, c_float, c_u64, c_u32, c_u32,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bbb_reparam_partials": (c_i64, [ctypes.POINTER(Segment), c_int]),
    "bbb_reparam_kl_bwd": (c_int, [ctypes.POINTER(Segment), c_int, c_int, c_float, c_float, c_u64, c_u32, c_u32,
                                   c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "bbb_adam_step": (c_int, [ctypes.POINTER(AdamSegment), c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                            c_i64, c_void_p, c_void_p, c_void_p]),
    "bbb_eps_dump": (c_int, [c_void_p, c_i64, c_i64, c_u64, c_u32, c_u32, c_void_p]),
    "bbb_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bbb_lrt_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_u64, c_u32, c_u32, c_int, c_void_p, c_void_p]),
    "bbb_conv2d_chwn_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bbb_conv2d_chwn_bf16x3_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_u32, c_void_p]),
    "bbb_maxpool_chwn_s3": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_s3_convert": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "bbb_conv2d_c8x3_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_u32, c_void_p]),
    "bbb_c8s3_convert": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_i64, c_int, c_int, c_void_p]),
    "bbb_w_tap_major": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "bbb_s2d_c8s3": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_s2d_c8s3sq": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_lrt_conv2d_c8x3_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_u64, c_u32,
                                        c_u32, c_int, c_void_p, c_u32, c_void_p]),
    "bbb_maxpool_chwn_s3sq": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_w_s2d_tap_major": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "bbb_lrt_conv2d_chwn_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_u64, c_u32, c_u32, c_int, c_void_p, c_void_p]),
    "bbb_lrt_sample_chwn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_u64, c_u32, c_u32, c_void_p,
                                  c_void_p]),
    "bbb_lrt_sample_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_u64, c_u32, c_u32, c_void_p, c_void_p]),
    "bbb_maxpool_chwn": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_pool_act_bwd_chwn": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_void_p]),
    "bbb_lrt_pool_act_bwd_chwn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    "bbb_conv2d_chwn_bf16_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_u32, c_void_p]),
    "bbb_maxpool_chwn_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_nchw_to_chwn_bf16": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_void_p]),
    "bbb_nchw_to_chwn_bf16_slices": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p]),
    "bbb_mc_tail": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bbb_mc_tail_cb": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bbb_mc_tail_cb_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_mc_tail_units": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bbb_mc_tail_units_step": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                       c_u32, c_void_p]),
    "bbb_mc_tail_groups_step": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                        c_u32, c_void_p]),
    "bbb_mc_tail_share_step": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                       c_u32, c_void_p]),
    "bbb_uncertainty": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bbb_transpose2d": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p]),
    "bbb_flip_transpose_w": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "bbb_flip_transpose_w_multi": (c_int, [ctypes.POINTER(FlipSeg), c_int, c_void_p]),
    "bbb_flip_transpose_w_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p]),
    "bbb_elbo_cb_fwd": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, ctypes.c_float, c_int, c_int, c_void_p, c_void_p]),
    "bbb_elbo_cb_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, ctypes.c_float, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_void_p]),
    "bbb_im2col_pbj": (c_int, [c_void_p, c_void_p, ctypes.POINTER(ConvDesc), c_void_p]),
    "bbb_transpose_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_void_p]),
    "bbb_transpose_sum_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_i32), ctypes.POINTER(c_i64),
                                          ctypes.POINTER(c_i64), c_i64, c_i64, c_int, c_i64, c_i64, c_void_p]),
    "bbb_conv2d_chwn_splitk_scratch": (c_i64, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(c_i32)]),
    "bbb_conv2d_chwn_splitk_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_i64,
                                           c_void_p]),
    "bbb_lrt_conv2d_chwn_splitk_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_u64, c_u32, c_u32, c_int, c_void_p, c_int, c_void_p, c_i64,
                                               c_void_p]),
    "bbb_plane_sum": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_void_p]),
    "bbb_sum_leading": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p]),
    "bbb_lrt_glue": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "bbb_abi_version": (c_int, []),
    "bbb_build_info": (ctypes.c_char_p, []),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class BBBHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the in-tree .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BBBHipError(
                f"{LIB_PATH} not found: the HIP kernels are not built. Run ./build.sh "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if h.bbb_abi_version() != ABI_VERSION:
            raise BBBHipError(f"ABI mismatch: library reports {h.bbb_abi_version()}, binding expects {ABI_VERSION}")
        _lib = h
    return _lib


_ERR = {-1: "BBB_EINVAL (bad argument)", -2: "BBB_EALIGN (pointer not 4-byte aligned)", -3: "BBB_ESHAPE (bad geometry)"}


def check(rc, what):
    if rc != 0:
        raise BBBHipError(f"{what} failed: {_ERR.get(rc, 'hipError_t %d' % rc)}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_device(*tensors, dtype=torch.float32):
    """Every compute entry point works on MI355X memory only; fail loudly otherwise."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise BBBHipError("bbb_hip computes on an MI355X only (tensor is on %s); there is no CPU path. "
                              "Move the module / input to 'cuda'." % t.device)
        if t.dtype != dtype:
            raise BBBHipError(f"bbb_hip expects {dtype} tensors here, got {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def cur_stream(device):
    """hipStream_t (as an integer) of torch's current stream on `device` -- asked before every launch, so through the raw
    accessor when this torch has it (torch.cuda.current_stream builds a Stream object each time: ~4 us)."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """Context that makes `device` the current HIP device for a launch; free when it already is (the usual case)."""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)
