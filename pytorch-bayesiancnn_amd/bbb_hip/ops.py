"""Tensor-level entry points over the C ABI + their autograd wiring.

Forward math is entirely in the HIP kernels.  Backward (training extension): the parameter pass
has its own fused HIP kernel (eps regenerated from the counter, never stored); conv / linear weight gradients and input
gradients run on the SAME fp32-MFMA implicit-GEMM kernel as the forward (wgrad = the axis-swapped convolution, dgrad = the
flipped-weight convolution of the stride-upsampled output gradient) -- no ATen / MIOpen / rocBLAS convolution or GEMM
anywhere.  Activations, pooling and the loss tail use torch autograd (element-wise ops) in this mode.
"""
import contextlib
import ctypes
import os
import itertools
import sys
import threading
import types
import weakref

import torch

from . import _lib, rng
from ._lib import Segment, ConvDesc, check, ptr, require_device, cur_stream, on_device

_scratch = {}
_retired = []     # (key, buffer) of outgrown scratch buffers: a captured hipGraph may have their address baked in, so a per-stream
                  # one is NEVER freed (a second pipeline / a larger model captured on the same stream must not pull the first
                  # graph's tickets or KL slots out from under it); a scope's go when its owner dies (_release_scope).  Sizes at
                  # least double, so a key retires < its final size in total.


_scope_tls = threading.local()
_scope_ids = itertools.count(1)


class scratch_scope:
    """Context: launches enqueued inside take their scratch buffers (KL partial slots, split-contraction tickets / partial tiles)
    from a set that belongs to `owner` instead of the per-(device, stream) set.  For captured graphs: the buffers' addresses are
    baked into the recorded launches, and a graph may be replayed on another stream than it was captured on (GraphedLogits) while
    torch's stream pool hands the capture stream's HANDLE to a later capture -- keyed by stream, the two graphs would then share
    tickets while replaying concurrently.  Keyed by owner, every graph has its own; the set is dropped when the owner dies."""

    def __init__(self, owner=None, token=None):
        if owner is None:                            # re-enter a scope by its token (autograd's device thread: fast_train backward)
            self.tok = token
            return
        tok = getattr(owner, "_bbb_scratch_token", None)
        if tok is None:
            tok = next(_scope_ids)
            try:
                owner._bbb_scratch_token = tok
                weakref.finalize(owner, _release_scope, tok)
            except (AttributeError, TypeError):
                # an owner that can carry neither the token nor a finaliser: a fresh token per call would leave a scratch set behind
                # every time and nothing would ever release it -- fall back to the per-(device, stream) set
                tok = None
        self.tok = tok

    def __enter__(self):
        self.prev = getattr(_scope_tls, "tok", None)
        _scope_tls.tok = self.tok
        return self

    def __exit__(self, *a):
        _scope_tls.tok = self.prev
        return False


def current_scratch_token():
    return getattr(_scope_tls, "tok", None)


def _release_scope(tok):
    mine = lambda k: isinstance(k[2], tuple) and k[2][:2] == ("scope", tok)
    for k in [k for k in _scratch if mine(k)]:
        del _scratch[k]
    _retired[:] = [(k, b) for k, b in _retired if not mine(k)]          # the scope's outgrown buffers: its graphs are gone too


def _scratch_key(device, kind):
    # (inside a scope still one set per stream: a captured step may fan its draws out over side streams that run concurrently)
    tok = getattr(_scope_tls, "tok", None)
    st = cur_stream(device)
    return (device.index, kind, ("scope", tok, st) if tok is not None else st)


def _grow(key, need, make):
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        if buf is not None:
            _retired.append((key, buf))
            need = max(int(need), 2 * buf.numel())
        buf = _scratch[key] = make(int(need))
    return buf


def _partials(device, n):
    # one scratch buffer per (device, stream): launches on different streams (graph lanes, a second model) may overlap.
    # every slot starts as "not published yet" (0xFF bytes); each launch re-arms the slots it consumed
    return _grow(_scratch_key(device, "kl"), max(int(n), 4096),
                 lambda m: torch.full((m,), -1, dtype=torch.int64, device=device).view(torch.float64))


def _segments(mus, rhos, ws, sigmas, epss, stream_ids, draws):
    arr = (Segment * len(mus))()
    for i, (mu, rho) in enumerate(zip(mus, rhos)):
        n = mu.numel()
        s = arr[i]
        s.mu, s.rho = mu.data_ptr(), rho.data_ptr()
        s.w = ptr(ws[i]) if ws is not None else 0
        s.sigma = ptr(sigmas[i]) if sigmas is not None else 0
        s.eps = ptr(epss[i]) if epss is not None else 0
        s.n = n
        s.draw_stride = n
        s.stream_id = stream_ids[i]
    return arr


def reparam_kl_forward(mus, rhos, prior_mu, prior_sigma, stream_ids, seed, call0, draws=1, sample=True,
                       want_sigma=False, sigma_squared=False, want_kl=True, eps=None, textbook_kl=False):
    """Raw (no autograd) fused pass.  mus/rhos: lists of contiguous fp32 device tensors.
    Returns (ws | None, sigmas | None, kl | None); ws[i] has shape [draws, *mus[i].shape]."""
    if len(mus) == 0 or len(mus) > _lib.MAX_SEGMENTS:
        raise _lib.BBBHipError(f"1..{_lib.MAX_SEGMENTS} tensors per launch, got {len(mus)}")
    require_device(*mus, *rhos)
    dev = mus[0].device
    mus = [m.contiguous() for m in mus]
    rhos = [r.contiguous() for r in rhos]
    ws = [torch.empty((draws,) + tuple(m.shape), dtype=torch.float32, device=dev) for m in mus] if sample else None
    sigmas = [torch.empty_like(m) for m in mus] if want_sigma else None
    if eps is not None:
        eps = [e.contiguous() for e in eps]
        require_device(*eps)
        for e, m in zip(eps, mus):
            if e.numel() != draws * m.numel():
                raise _lib.BBBHipError("external eps must have shape [draws, *param.shape]")
    segs = _segments(mus, rhos, ws, sigmas, eps, stream_ids, draws)
    kl = torch.empty((), dtype=torch.float32, device=dev) if want_kl else None
    L = _lib.lib()
    nparts = L.bbb_reparam_partials(segs, len(mus))
    parts = _partials(dev, nparts) if want_kl else None
    flags = (_lib.SIGMA_SQUARED if sigma_squared else 0) | (_lib.KL_TEXTBOOK if textbook_kl else 0)
    with on_device(dev):
        rc = L.bbb_reparam_kl_fwd(segs, len(mus), draws, float(prior_mu), float(prior_sigma), seed, call0 & 0xFFFFFFFF,
                                  flags, ptr(parts), ptr(kl), 0, rng.call_dev_ptr(dev), cur_stream(dev))
    check(rc, "bbb_reparam_kl_fwd")
    return ws, sigmas, kl


def reparam_kl_backward(mus, rhos, gws, gkl, prior_mu, prior_sigma, stream_ids, seed, call0, draws, eps=None,
                        textbook_kl=False, gsigmas=None, sigma_squared=False, gws_mean_only=False):
    """grad_mu[i], grad_rho[i] for every tensor (see bbb_reparam_kl_bwd).  gsigmas: gradients w.r.t. the forward's sigma
    (sigma_squared: sigma^2) outputs, entries may be None.  gws_mean_only: gws are gradients w.r.t. the means themselves
    (BBB_GW_MEAN_ONLY: added into grad_mu, nothing into grad_rho)."""
    require_device(*mus, *rhos)
    dev = mus[0].device
    mus = [m.contiguous() for m in mus]
    rhos = [r.contiguous() for r in rhos]
    gws = [None if g is None else g.contiguous() for g in gws]
    if eps is not None:
        eps = [e.contiguous() for e in eps]
    if gsigmas is not None:
        gsigmas = [None if g is None else g.to(dtype=torch.float32).contiguous() for g in gsigmas]
        require_device(*gsigmas)
    segs = _segments(mus, rhos, gws, gsigmas, eps, stream_ids, draws)
    gmu = [torch.empty_like(m) for m in mus]
    grho = [torch.empty_like(m) for m in mus]
    pm = (ctypes.c_void_p * len(mus))(*[g.data_ptr() for g in gmu])
    pr = (ctypes.c_void_p * len(mus))(*[g.data_ptr() for g in grho])
    if gkl is not None:
        gkl = gkl.to(device=dev, dtype=torch.float32).contiguous()
    flags = (_lib.KL_TEXTBOOK if textbook_kl else 0) | (_lib.SIGMA_SQUARED if (sigma_squared and gsigmas is not None) else 0)
    flags |= _lib.GW_MEAN_ONLY if gws_mean_only else 0
    with on_device(dev):
        rc = _lib.lib().bbb_reparam_kl_bwd(segs, len(mus), draws, float(prior_mu), float(prior_sigma), seed,
                                           call0 & 0xFFFFFFFF, flags, ptr(gkl), pm, pr, rng.call_dev_ptr(dev), cur_stream(dev))
    check(rc, "bbb_reparam_kl_bwd")
    return gmu, grho


def eps_dump(n, seed, call, stream_id, device, start=0):
    out = torch.empty(n, dtype=torch.float32, device=device)
    with on_device(out.device):
        check(_lib.lib().bbb_eps_dump(out.data_ptr(), n, start, seed, call & 0xFFFFFFFF, stream_id, cur_stream(out.device)),
              "bbb_eps_dump")
    return out


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


ACT_CODE = {None: 0, "none": 0, "relu": 1, "softplus": 2}


class _Shape:
    """A stand-in for a tensor where only `.shape` is read (_desc): building a meta tensor costs ~3 us of host time per launch."""
    __slots__ = ("shape",)

    def __init__(self, shape):
        self.shape = tuple(shape)


def _desc(x, w, stride, padding, dilation, draws, x_shared, w_shared, act):
    """x: [E|1... folded][B, Cin, H, W] given as 5-d [Ex, B, Cin, H, W]; w: [Ew, Cout, Cin, kh, kw]."""
    d = ConvDesc()
    Ex, B, Cin, H, W = x.shape
    Ew, Cout, Cin2, kh, kw = w.shape
    if Cin != Cin2:
        raise _lib.BBBHipError(f"channel mismatch: input has {Cin}, weight expects {Cin2}")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    d.batch, d.cin, d.h, d.w, d.cout, d.kh, d.kw = B, Cin, H, W, Cout, kh, kw
    d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.dil_h, d.dil_w = sh, sw, ph, pw, dh, dw
    d.draws = draws
    d.x_draw_stride = 0 if x_shared else B * Cin * H * W
    d.w_draw_stride = 0 if w_shared else Cout * Cin * kh * kw
    d.b_draw_stride = 0 if w_shared else Cout
    d.act = ACT_CODE[act]
    ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    return d, ho, wo


def conv2d_forward(x, w, bias, stride=1, padding=0, dilation=1, act=None):
    """Batched-over-draws conv on the fp32 matrix cores (no autograd).
    x: [E, B, Cin, H, W] or [1, B, Cin, H, W] (shared by all draws); w: [E, Cout, Cin, kh, kw] (or [1, ...]
    shared); bias: [E|1, Cout] or None.  Returns y [E, B, Cout, Ho, Wo] with E = max of the leading dims."""
    require_device(x, w, bias)
    x, w = x.contiguous(), w.contiguous()
    bias = None if bias is None else bias.contiguous()
    E = max(x.shape[0], w.shape[0])
    if x.shape[0] not in (1, E) or w.shape[0] not in (1, E):
        raise _lib.BBBHipError("leading (draw) dims of x and w must be 1 or equal")
    d, ho, wo = _desc(x, w, stride, padding, dilation, E, x.shape[0] == 1 and E > 1, w.shape[0] == 1 and E > 1, act)
    y = torch.empty((E, x.shape[1], w.shape[1], ho, wo), dtype=torch.float32, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), ptr(bias), y.data_ptr(),
                                        cur_stream(x.device)), "bbb_conv2d_fwd")
    return y


def lrt_conv2d_forward(x, w_mu, w_var, b_mu, b_var, seed, call0, stream_id, stride=1, padding=0, dilation=1,
                       sample=True, eps=None, want_moments=False, act=None):
    """LRT layer in one launch.  x: [E, B, Cin, H, W] (E independent sample slabs); weights shared.
    Returns (y, act_mu | None, act_var | None), each [E, B, Cout, Ho, Wo]."""
    require_device(x, w_mu, w_var, b_mu, b_var, eps)
    x = x.contiguous()
    w_mu, w_var = w_mu.contiguous(), w_var.contiguous()
    b_mu = None if b_mu is None else b_mu.contiguous()
    b_var = None if b_var is None else b_var.contiguous()
    E = x.shape[0]
    d, ho, wo = _desc(x, w_mu.unsqueeze(0), stride, padding, dilation, E, False, True, act)
    d.w_draw_stride = 0
    d.b_draw_stride = 0
    shape = (E, x.shape[1], w_mu.shape[0], ho, wo)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    am = torch.empty(shape, dtype=torch.float32, device=x.device) if want_moments else None
    av = torch.empty(shape, dtype=torch.float32, device=x.device) if want_moments else None
    if eps is not None:
        eps = eps.contiguous()
        if eps.numel() != y.numel():
            raise _lib.BBBHipError("external eps must have the output's shape")
    with on_device(x.device):
        check(_lib.lib().bbb_lrt_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w_mu.data_ptr(), w_var.data_ptr(), ptr(b_mu),
                                            ptr(b_var), y.data_ptr(), ptr(am), ptr(av), ptr(eps), seed,
                                            call0 & 0xFFFFFFFF, stream_id, 1 if sample else 0, rng.call_dev_ptr(x.device),
                                            cur_stream(x.device)),
              "bbb_lrt_conv2d_fwd")
    return y, am, av


@contextlib.contextmanager
def graph_capture(graph, stream=None):
    """torch.cuda.graph(graph, stream, thread-local error mode) with the cyclic garbage collector out of the way: this torch no
    longer collects before a capture, and a dead reference cycle that owns an older hipGraph (or any device memory) may be
    collected at ANY allocation -- inside the capture, where destroying a graph is an illegal HIP call that surfaces as
    `operation not permitted when stream is capturing` from a destructor, i.e. terminate().  Collect first, keep the collector off
    until the capture has ended."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            yield
    finally:
        if was:
            gc.enable()


class LaunchConfig:
    """The launch-shape / arithmetic-mode choices of the batch-innermost path, as ONE object instead of process-wide switches.

    A `GraphedMC` / `GraphedPipeline` / `GraphedLogits` / `GraphedTrainStep` snapshots the configuration that is current when it
    is built (or takes `launch_config=`) and enqueues every launch under it, so two models -- or two pipelines of one model --
    with different modes coexist in one process and on several threads.  `ops.gemm_mode = ...` and friends still exist: they
    read / write the fields of the process DEFAULT configuration (what applies when nobody asked for anything else).

    gemm_mode   "fp32" | "bf16x3": the batch-innermost BBB GEMM launches (inference AND the role-swapped gradient launches of the
                training path) run their contraction on the 16-bit matrix pipe at fp32 accuracy (bbb_conv2d_chwn_bf16x3_fwd:
                every operand element split into three bf16 pieces while staged, six products, fp32 accumulation).  Range-free
                (bf16 has fp32's exponent range: no operand windows or scales).  Opt-in: results agree with the fp32 kernel
                to rounding, not bit for bit.  LRT layers keep the fused fp32 kernel (one staged x tile feeds both of its
                contractions; a split form would need twelve operand planes in LDS).
    bf16x3_min_workgroups   smaller launches stay on the fp32 kernel (and the layer's split contraction): measured faster there
    s3_min_images   split-bf16 mode: steps of at least this many (draw x image) rows keep their activations in the split format S3
                between layers (ensemble._mc_logits_chwn); smaller steps split while staging, per launch
    split_k     layers with few (pixel, channel-tile) groups and a long contraction (AlexNet conv4 / conv5) add their k ranges'
                partial sums in range order (bbb_conv2d_chwn_splitk_fwd): a property of the LAYER, identical for every launch
                size and partition.  False: the plain fmaf chain everywhere (bit-identical to the reference-layout kernel).
    pool_fusion a conv layer followed by [activation ->] MaxPool2d(2, 2) may run as ONE launch (bbb_conv_desc_t::pool)
    pool_fuse_min_items / pool_fuse_imbalance   a launch with nothing else in flight: only when it still has this many (pooled pixel,
                64-channel tile, 128-image tile, draw) items and they spread over the 256 CUs to within this factor (4x longer items)
    launches_overlap   the caller enqueues work that runs beside other lanes' kernels: the under-filled tail of a fused launch is
                then filled by them, and fusing pays from ~2 items per CU on (measured, profiles/r04_notes.md section 7: -2 to
                -3 % per step at 10 draws x 3 lanes, +4.5 % with one lane; at 5 draws per launch it no longer does)
    pool_fuse_weight_budget   bytes of weight tiles concurrently live per XCD that a fused launch may have (see pool_fusion_ok)
    c8x3        split-bf16 mode: layers with Cin % 16 == 0 take the MFMA-ready-operand kernel (csrc/pconv_c8x3.hip), whatever the
                launch size (a property of the layer: partitions of a step keep its bits)
    c8x3_s2d    ... and a strided first layer on few channels (s2d_layer_ok: AlexNet conv1) joins them in space-to-depth form
    dropin_precision   "fp32" | "bf16": what a whole-model `net(x)` through the drop-in layers computes in.  "bf16" = the bf16 storage
                path of ensemble.mc_forward(precision="bf16") (sampled weights and activations in bf16, fp32 accumulation, fp32 logits)
                behind the unmodified `for j in range(num_ens): net(x)` loop -- inference only, BBB models the batch-innermost path
                covers; anything else (autograd enabled, LRT layers, hooks, a layer called on its own) raises instead of silently
                computing in fp32"""
    FIELDS = ("gemm_mode", "bf16x3_min_workgroups", "s3_min_images", "split_k", "pool_fusion", "pool_fuse_min_items",
              "pool_fuse_imbalance", "launches_overlap", "pool_fuse_min_items_overlapped", "pool_fuse_weight_budget",
              "bf16_pool_fuse_min_rows", "bf16_pool_fuse_min_rows_overlapped", "bf16_c8", "bf16_c8_min_items", "c8x3", "c8x3_s2d",
              "dropin_precision")
    __slots__ = FIELDS

    def __init__(self, **kw):
        self.gemm_mode = "fp32"
        self.bf16x3_min_workgroups = 256
        self.s3_min_images = 2048
        self.split_k = True
        self.pool_fusion = True
        self.pool_fuse_min_items = 1536
        self.pool_fuse_imbalance = 1.07
        self.launches_overlap = False
        self.pool_fuse_min_items_overlapped = 400
        self.pool_fuse_weight_budget = 1 << 20
        self.bf16_pool_fuse_min_rows = 0               # (bf16_pool_fusion_ok) pooled rows x image tiles x slabs from which the pooled
        self.bf16_pool_fuse_min_rows_overlapped = 0    # first-layer form of the bf16 path is taken (0: always -- the library's
                                                       # window-resident form wins from one step per launch on); ... beside other lanes
        self.bf16_c8 = True                            # (bf16_c8_input_ok) channel-interleaved activations between a pooled first layer and a
        self.bf16_c8_min_items = 0                     # layer that has the strip form over them, for launches of at least this many strip
                                                       # workgroups (0: always -- measured faster from one step per launch on); False: never
        self.dropin_precision = "fp32"
        self.c8x3 = True                               # split-bf16 mode: layers (behind the first) with Cin % 32 == 0 run on the
                                                       # MFMA-ready-operand kernel (conv2d_c8x3_forward: channel-interleaved split
                                                       # activations + tap-major weights from the parameter pass); False: round 4's
                                                       # kernel (split while staging / planar S3) everywhere
        self.c8x3_s2d = True                           # ... and a strided first layer on few channels joins that chain in space-to-depth form
                                                       # (s2d_layer_ok: AlexNet conv1), pooling inside its launch; False: fp32 kernel
        for k, v in kw.items():
            setattr(self, k, v)              # (unknown names raise: __slots__)

    def copy(self, **kw):
        c = LaunchConfig(**{k: getattr(self, k) for k in self.FIELDS})
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def key(self):
        """Hashable value of every field (cache keys: a result / a captured graph depends on all of them)."""
        return tuple(getattr(self, k) for k in self.FIELDS)

    def __repr__(self):
        return "LaunchConfig(%s)" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in self.FIELDS)


_default_config = LaunchConfig()
_tls = threading.local()


def current_config():
    """The configuration launches are enqueued under right now: the innermost `use_config` of THIS thread, else the process default."""
    st = getattr(_tls, "stack", None)
    return st[-1] if st else _default_config


class use_config:
    """Context: enqueue under `cfg` (a LaunchConfig), or under the current configuration with some fields replaced
    (`use_config(gemm_mode="bf16x3")`).  Thread-local, re-entrant."""

    def __init__(self, cfg=None, **fields):
        self.cfg, self.fields = cfg, fields

    def __enter__(self):
        base = self.cfg if self.cfg is not None else current_config()
        cfg = base.copy(**self.fields) if self.fields else base
        st = getattr(_tls, "stack", None)
        if st is None:
            st = _tls.stack = []
        st.append(cfg)
        return cfg

    def __exit__(self, *a):
        _tls.stack.pop()
        return False


_split_plans = {}


def _split_scratch(d, lrt, device):
    """(k_split, scratch tensor | None) for this launch.  k_split is the LAYER's plan (a function of its geometry only:
    bbb_conv2d_chwn_splitk_scratch), applied to every launch of the layer whatever its size, so that a draw computed alone, in
    a 10-draw launch, as a work unit or as one of several steps per launch is the same number bit for bit; the scratch (a
    zero-initialised per-(device, stream) buffer whose arrival tickets stay zero from launch to launch) is only there when
    this launch is small enough for the cross-workgroup form -- larger launches run the same summation order inside one
    workgroup per tile."""
    if not current_config().split_k:
        return 1, None
    pkey = (bytes(d), bool(lrt))
    plan = _split_plans.get(pkey)
    if plan is None:
        ks = ctypes.c_int32(1)
        need = _lib.lib().bbb_conv2d_chwn_splitk_scratch(ctypes.byref(d), 1 if lrt else 0, ctypes.byref(ks))
        plan = _split_plans[pkey] = (ks.value, int(need))
    ks_v, need = plan
    if ks_v <= 1:
        return 1, None
    if need <= 0:
        return ks_v, None
    buf = _grow(_scratch_key(device, "splitk"), max(int(need), 1 << 22),
                lambda m: torch.zeros(m, dtype=torch.uint8, device=device))
    return ks_v, buf


def _apply_units(d, units, x_per_slice):
    """units = (S, off) or None: the launch's slabs are work units u = off + e of the draw-major (draw, batch slice) grid
    (see bbb_conv_desc_t).  x_per_slice: the input is an [S][...] per-slice tensor shared by all draws (first layer)."""
    if units is None:
        return
    S, off = int(units[0]), int(units[1])
    if S > 1:
        d.unit_div, d.unit_off = S, off % S
        d.x_unit_mod = S if x_per_slice else 0


def _desc_chwn(x, w, stride, padding, dilation, draws, x_shared, w_shared, act):
    """x: [Ex, Cin, H, W, B]; w: [Ew, Cout, Cin, kh, kw]."""
    Ex, Cin, H, W, B = x.shape
    d, ho, wo = _desc(_Shape((Ex, B, Cin, H, W)), w, stride, padding, dilation, draws, x_shared, w_shared, act)
    d.x_draw_stride = 0 if x_shared else Cin * H * W * B
    return d, ho, wo


class overlapped_launches(use_config):
    """Context: the launches enqueued inside run concurrently with other streams' kernels (a lane of a GraphedPipeline)."""

    def __init__(self, on=True):
        super().__init__(launches_overlap=bool(on))


def pool_fusion_ok(x_shape, w_shape, stride, padding, dilation, draws, pool_module=None, fp32_kernel=False):
    """Should conv2d_chwn_forward(..., pool=True) replace conv + maxpool_chwn(2, 2) for this launch?  Same bits either way; the
    fused launch saves the pooling launch and 3/4 of the layer's output traffic, but its items are four times fewer and four
    times longer: chosen when other lanes' kernels run beside it (launches_overlap), else only when its items still fill the chip
    evenly (measured: profiles/r04_notes.md section 7).
    x_shape [*, Cin, H, W, B], w_shape [*, Cout, Cin, kh, kw]; pool_module: the nn.MaxPool2d that follows (checked for 2 / 2)."""
    cfg = current_config()
    if not cfg.pool_fusion or (cfg.gemm_mode != "fp32" and not fp32_kernel):       # (fp32_kernel: the caller runs THIS launch on the fp32 kernel)
        return False
    key = (tuple(x_shape[-4:]), tuple(w_shape[-4:]), _pair(stride), _pair(padding), _pair(dilation), int(draws), cfg.launches_overlap,
           cfg.pool_fuse_min_items, cfg.pool_fuse_imbalance, cfg.pool_fuse_min_items_overlapped, cfg.pool_fuse_weight_budget,
           None if pool_module is None else (str(pool_module.kernel_size), str(pool_module.stride), str(pool_module.padding),
                                             str(pool_module.dilation), pool_module.ceil_mode,
                                             getattr(pool_module, "return_indices", False)))
    hit = _pool_rule_cache.get(key)
    if hit is None:
        if len(_pool_rule_cache) > 512:
            _pool_rule_cache.clear()
        hit = _pool_rule_cache[key] = _pool_fusion_rule(x_shape, w_shape, stride, padding, dilation, draws, pool_module)
    return hit


_pool_rule_cache = {}


def is_pool_2x2(pool_module):
    """Is this nn.MaxPool2d the plain 2 x 2 / stride 2 window (what the fused launches implement)?"""
    pr = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    return not (pr(pool_module.kernel_size) != (2, 2) or pr(pool_module.stride if pool_module.stride is not None else 2) != (2, 2) or
                pr(pool_module.padding) != (0, 0) or pr(pool_module.dilation) != (1, 1) or pool_module.ceil_mode or
                getattr(pool_module, "return_indices", False))


def _pool_fusion_rule(x_shape, w_shape, stride, padding, dilation, draws, pool_module):
    cfg = current_config()
    if pool_module is not None and not is_pool_2x2(pool_module):
        return False
    B = x_shape[-1]
    xm = torch.empty((1,) + tuple(x_shape[-4:]), device="meta")
    wm = torch.empty((1,) + tuple(w_shape[-4:]), device="meta")
    d, ho, wo = _desc_chwn(xm, wm, stride, padding, dilation, int(draws), False, False, None)
    if ho % 2 or wo % 2 or B % 4:
        return False
    ks = ctypes.c_int32(1)
    _lib.lib().bbb_conv2d_chwn_splitk_scratch(ctypes.byref(d), 0, ctypes.byref(ks))
    if ks.value > 1:
        return False
    # the weight tiles live in one XCD's L2 at a time: an XCD runs 128 workgroups = 128 / (pooled pixels x image tiles) groups of
    # items that share a (draw, 64-channel) weight tile -- four times the unfused launch's, whose groups hold four times the items.
    # AlexNet conv2 on CIFAR maps (4 pooled pixels x 4 tiles: 8 tiles of 410 KB) overflowed the 4 MB L2: 404 MB fetched per step for
    # 25 MB of operands, and no time gained (profiles/r04_notes.md section 7)
    per_group = (ho // 2) * (wo // 2) * -(-B // 128)
    if max(1, 128 // per_group) * 64 * w_shape[-3] * w_shape[-2] * w_shape[-1] * 4 > cfg.pool_fuse_weight_budget:
        return False
    items = int(draws) * per_group * -(-w_shape[-4] // 64)
    if cfg.launches_overlap:
        return items >= cfg.pool_fuse_min_items_overlapped
    return items >= cfg.pool_fuse_min_items and -(-items // 256) * 256 <= cfg.pool_fuse_imbalance * items


def conv2d_chwn_forward(x, w, bias, stride=1, padding=0, dilation=1, act=None, out=None, units=None, n_units=None,
                        x_per_slice=False, x_div=1, bf16x3=None, x_s3=False, out_s3=False, x_off=0, pool=False, w_tap_major=False):
    """Batch-innermost conv for the ensemble path.  x: [E|1, Cin, H, W, B] (B % 4 == 0); w: [E|1, Cout, Cin, kh, kw];
    bias [E|1, Cout] or None -> y [E, Cout, Ho, Wo, B].  Padding taps are skipped, not multiplied.
    Work units (ensemble sharding): units = (S, off), n_units = U output slabs; w / bias hold the weight sets of the draws
    the units touch, x is [U, ...] or, for a layer whose input is the same for every draw, the per-slice [S, Cin, H, W, Bs].
    x_div = D > 1 (several Monte-Carlo steps per launch): x holds E / D input slabs and output slab e reads slab e // D
    (bbb_conv_desc_t::x_unit_div) -- the first layer of G steps x D draws, each step on its own batch.  x_off (< D): output slab
    e reads slab (e + x_off) // D -- a rank's share of a group of steps that starts in the middle of a step.
    bf16x3: True / False = run this launch on the split-bf16 kernel or not; None (default) = ops.gemm_mode decides.
    x_s3 / out_s3 (split-bf16 kernel only, B % 8 == 0): the input / output travels in the split activation format S3 -- a bf16
    tensor [E|1, 3, C, H, W, B] holding the hi / mid / lo pieces of the same fp32 values (s3_from_f32 / s3_to_f32).
    pool = True (fp32 kernel, layers without a split contraction, even Ho and Wo): the launch also applies MaxPool2d(2, 2) to the
    activated output -> y [E, Cout, Ho/2, Wo/2, B], bit for bit maxpool_chwn(conv2d_chwn_forward(...), 2, 2) (pool_fusion_ok says
    when that pays).
    w_tap_major = True (fp32 kernel only): w is given as [E|1, Cout, kh, kw, Cin] and the contraction runs tap-major
    (bbb_conv_desc_t::w_tap_major) -- the same products in another summation order."""
    require_device(w, bias)
    require_device(x, dtype=torch.bfloat16 if x_s3 else torch.float32)
    x, w = x.contiguous(), w.contiguous()
    bias = None if bias is None else bias.contiguous()
    w_mem = w
    if w_tap_major:
        if x_s3 or out_s3 or bf16x3:
            raise _lib.BBBHipError("w_tap_major: fp32 kernel only")
        bf16x3 = False
        w = _Shape((w.shape[0], w.shape[1], w.shape[4], w.shape[2], w.shape[3]))
    if x_s3:
        if x.dim() != 6 or x.shape[1] != 3 or x.shape[5] % 8:
            raise _lib.BBBHipError("an S3 input is a bf16 tensor [E|1, 3, C, H, W, B] with B % 8 == 0")
        x5 = _Shape((x.shape[0],) + tuple(x.shape[2:]))
    else:
        x5 = x
    if units is not None and units[0] > 1:
        E = int(n_units)
        if x5.shape[0] != (units[0] if x_per_slice else E):
            raise _lib.BBBHipError("work units: x must hold one slab per unit, or one per batch slice with x_per_slice")
        d, ho, wo = _desc_chwn(x5, w, stride, padding, dilation, E, False, False, act)
        _apply_units(d, units, x_per_slice)
    elif int(x_div) > 1:
        E = w.shape[0]
        if not 0 <= int(x_off) < int(x_div) or x5.shape[0] != -(-(E + int(x_off)) // int(x_div)):
            raise _lib.BBBHipError("x_div: x must hold ceil((E + x_off) / x_div) input slabs for the E weight sets")
        d, ho, wo = _desc_chwn(x5, w, stride, padding, dilation, E, False, False, act)
        d.x_unit_div, d.x_unit_off = int(x_div), int(x_off)
    else:
        E = max(x5.shape[0], w.shape[0])
        if x5.shape[0] not in (1, E) or w.shape[0] not in (1, E):
            raise _lib.BBBHipError("leading (draw) dims of x and w must be 1 or equal")
        d, ho, wo = _desc_chwn(x5, w, stride, padding, dilation, E, x5.shape[0] == 1 and E > 1, w.shape[0] == 1 and E > 1, act)
    if x_s3:
        d.x_draw_stride *= 3                                   # bf16 elements per S3 slab
    if w_tap_major:
        d.w_tap_major = 1
    B = x5.shape[4]
    if pool:
        if x_s3 or out_s3 or bf16x3 or ho % 2 or wo % 2:
            raise _lib.BBBHipError("pool=True: fp32 kernel only, even output height and width")
        d.pool = 1
        ho, wo = ho // 2, wo // 2
    shape = (E, 3, w.shape[1], ho, wo, B) if out_s3 else (E, w.shape[1], ho, wo, B)
    odt = torch.bfloat16 if out_s3 else torch.float32
    if out_s3 and B % 8:
        raise _lib.BBBHipError("an S3 output needs B % 8 == 0")
    if out is None:
        y = torch.empty(shape, dtype=odt, device=x.device)
    else:
        if out.numel() != E * (3 if out_s3 else 1) * w.shape[1] * ho * wo * B or not out.is_contiguous() or out.dtype != odt:
            raise _lib.BBBHipError("out= must be a contiguous tensor of the output's size and dtype")
        y = out.view(shape)
    with on_device(x.device):
        if x_s3 or out_s3 or (not pool and (bf16x3 if bf16x3 is not None else current_config().gemm_mode == "bf16x3")
                              and E * ho * wo * -(-w.shape[1] // 64) * -(-B // 128) >= current_config().bf16x3_min_workgroups):
            check(_lib.lib().bbb_conv2d_chwn_bf16x3_fwd(ctypes.byref(d), x.data_ptr(), w_mem.data_ptr(), ptr(bias), y.data_ptr(),
                                                        (1 if x_s3 else 0) | (2 if out_s3 else 0), cur_stream(x.device)),
                  "bbb_conv2d_chwn_bf16x3_fwd")
            return y
        ks, scr = _split_scratch(d, False, x.device)
        if ks > 1 and pool:
            raise _lib.BBBHipError("pool=True: this layer's contraction is split (conv + maxpool_chwn instead)")
        if ks > 1:
            check(_lib.lib().bbb_conv2d_chwn_splitk_fwd(ctypes.byref(d), x.data_ptr(), w_mem.data_ptr(), ptr(bias), y.data_ptr(), ks,
                                                        ptr(scr), 0 if scr is None else scr.numel(), cur_stream(x.device)),
                  "bbb_conv2d_chwn_splitk_fwd")
        else:
            check(_lib.lib().bbb_conv2d_chwn_fwd(ctypes.byref(d), x.data_ptr(), w_mem.data_ptr(), ptr(bias), y.data_ptr(),
                                                 cur_stream(x.device)), "bbb_conv2d_chwn_fwd")
    return y


def s3_from_f32(x):
    """fp32 [E, ...] -> its split activation form S3: bf16 [E, 3, ...] (hi, mid, lo pieces; exact).  The trailing dims must hold
    a multiple of 8 elements."""
    require_device(x)
    x = x.contiguous()
    n = x.numel() // x.shape[0]
    y = torch.empty((x.shape[0], 3) + tuple(x.shape[1:]), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_s3_convert(x.data_ptr(), y.data_ptr(), x.shape[0], n, 1, cur_stream(x.device)), "bbb_s3_convert")
    return y


def s3_to_f32(x):
    """S3 bf16 [E, 3, ...] -> the fp32 tensor [E, ...] it stores (hi + mid + lo, exact)."""
    require_device(x, dtype=torch.bfloat16)
    x = x.contiguous()
    n = x.numel() // (3 * x.shape[0])
    y = torch.empty((x.shape[0],) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_s3_convert(x.data_ptr(), y.data_ptr(), x.shape[0], n, 0, cur_stream(x.device)), "bbb_s3_convert")
    return y


def maxpool_chwn_s3(x, k, s):
    """MaxPool2d(k, s) on an S3 tensor [E, 3, C, H, W, B] (B % 8 == 0) -> [E, 3, C, Ho, Wo, B]."""
    require_device(x, dtype=torch.bfloat16)
    x = x.contiguous()
    E, three, C, H, W, B = x.shape
    ho, wo = (H - k) // s + 1, (W - k) // s + 1
    y = torch.empty((E, 3, C, ho, wo, B), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_maxpool_chwn_s3(x.data_ptr(), y.data_ptr(), E, C, H, W, B, int(k), int(s), cur_stream(x.device)),
              "bbb_maxpool_chwn_s3")
    return y


# ---- split-bf16 contraction over MFMA-ready operands (csrc/pconv_c8x3.hip): channel-interleaved split activations + tap-major weights
def c8s3_from_f32(x, squares=False):
    """fp32 batch-innermost [E, C, H, W, B] (C % 8 == 0) -> "c8 S3": bf16 [E, 3, C / 8, H, W, B, 8], the hi / mid / lo pieces
    (exact: hi + mid + lo is the fp32 value) of the 8 channels 8g .. 8g + 7 of an image adjacent in memory.
    squares = True: SIX planes [E, 6, ...] -- the pieces of the values, then the pieces of their squares (fp32 products): the slabs
    of the LRT chain (lrt_conv2d_c8x3_forward's second contraction reads the squares)."""
    require_device(x)
    x = x.contiguous()
    E, C, H, W, B = x.shape
    if C % 8:
        raise _lib.BBBHipError("c8 S3 needs a multiple of 8 channels")
    y = torch.empty((E, 6 if squares else 3, C // 8, H, W, B, 8), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_c8s3_convert(x.data_ptr(), y.data_ptr(), E, C, H * W, B, 2 if squares else 1, cur_stream(x.device)), "bbb_c8s3_convert")
    return y


def c8s3_to_f32(x):
    """c8 S3 [E, 3 | 6, C / 8, H, W, B, 8] -> the fp32 batch-innermost tensor [E, C, H, W, B] it stores (exact; six-plane slabs: their
    values)."""
    require_device(x, dtype=torch.bfloat16)
    x = x.contiguous()
    E, planes, CG, H, W, B, eight = x.shape
    y = torch.empty((E, CG * 8, H, W, B), dtype=torch.float32, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_c8s3_convert(x.data_ptr(), y.data_ptr(), E, CG * 8, H * W, B, 3 if planes == 6 else 0, cur_stream(x.device)),
              "bbb_c8s3_convert")
    return y


def w_tap_major(w):
    """[E, Cout, Cin, kh, kw] -> the tap-major rows [E, Cout, kh * kw, Cin] (what the parameter pass writes directly for a
    segment with w_tm_cin set: sample_weights_tm)."""
    require_device(w)
    w = w.contiguous()
    E, Cout, Cin, kh, kw = w.shape
    y = torch.empty((E, Cout, kh * kw, Cin), dtype=torch.float32, device=w.device)
    with on_device(w.device):
        check(_lib.lib().bbb_w_tap_major(w.data_ptr(), y.data_ptr(), E * Cout, Cin, kh * kw, cur_stream(w.device)), "bbb_w_tap_major")
    return y


def s2d_geometry(cin, kernel_size, stride, padding, dilation, h, w):
    """Space-to-depth form of a strided square layer (bbb_s2d_c8s3 / bbb_w_s2d_tap_major): None, or (m, C', Hb, Wb, Ho, Wo) -- the
    layer as an m x m convolution, stride 1, no padding, over a block image of C' channels and Hb x Wb positions."""
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    if kh != kw or sh != sw or ph != pw or (dh, dw) != (1, 1):
        return None
    ho, wo = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    if ho <= 0 or wo <= 0:
        return None
    m = -(-kh // sh)
    cp = -(-(cin * sh * sh) // 16) * 16
    return m, cp, ho + m - 1, wo + m - 1, ho, wo


def s2d_zero_border(cin, kernel_size, stride, padding, h, w):
    """Block rows / columns of the layer's space-to-depth image that lie entirely in the padding: (lead_h, lead_w, trail_h, trail_w)."""
    g = s2d_geometry(cin, kernel_size, stride, padding, 1, h, w)
    s, p = _pair(stride)[0], _pair(padding)[0]
    return (p // s, p // s, max(0, g[2] - -(-(p + h) // s)), max(0, g[3] - -(-(p + w) // s)))


def s2d_layer_ok(cin, cout, kernel_size, stride, padding, dilation, h, w, max_waste=1.5):
    """Should a layer that bbb_conv2d_c8x3_fwd cannot take directly (few input channels) run on it in space-to-depth form?  Yes when
    the block form's contraction (m * m * C' products per output) is at most max_waste times the layer's own (kh * kw * Cin): AlexNet
    conv1 (3 channels, 11 x 11, stride 4: 9 x 48 = 432 against 363, 1.19); not a 5 x 5 stride-1 layer on 3 channels (25 x 16 against
    75)."""
    g = s2d_geometry(cin, kernel_size, stride, padding, dilation, h, w)
    if g is None or cout % 8:
        return False
    kh, kw = _pair(kernel_size)
    return g[0] * g[0] * g[1] <= max_waste * kh * kw * cin


def s2d_c8s3(x, blocks, kernel_size, stride, padding, squares=False):
    """The caller's NCHW fp32 batch [blocks * Bs, C, H, W] -> the c8 S3 block image [blocks, 3, C' / 8, Hb, Wb, Bs, 8] of the layer
    (kernel_size, stride, padding): see s2d_geometry.  squares = True: six planes (the squares' pieces behind: an LRT first layer)."""
    require_device(x)
    x = x.contiguous()
    N, C, H, W = x.shape
    k, st, pd = _pair(kernel_size)[0], _pair(stride)[0], _pair(padding)[0]
    g = s2d_geometry(C, kernel_size, stride, padding, 1, H, W)
    if g is None or N % blocks:
        raise _lib.BBBHipError("space-to-depth form: square kernel / stride / padding, no dilation, the batch a multiple of `blocks`")
    m, cp, hb, wb, _, _ = g
    y = torch.empty((blocks, 6 if squares else 3, cp // 8, hb, wb, N // blocks, 8), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        fn = _lib.lib().bbb_s2d_c8s3sq if squares else _lib.lib().bbb_s2d_c8s3
        check(fn(x.data_ptr(), y.data_ptr(), blocks, N // blocks, C, H, W, k, st, pd, cur_stream(x.device)), "bbb_s2d_c8s3")
    return y


def w_s2d_tap_major(w, stride):
    """[E, Cout, Cin, k, k] -> the tap-major rows [E, Cout, m * m, C'] of the space-to-depth layer (zeros outside the k x k taps and in
    the padding channels)."""
    require_device(w)
    w = w.contiguous()
    E, Cout, Cin, kh, kw = w.shape
    st = _pair(stride)[0]
    m = -(-kh // st)
    cp = -(-(Cin * st * st) // 16) * 16
    y = torch.empty((E, Cout, m * m, cp), dtype=torch.float32, device=w.device)
    with on_device(w.device):
        check(_lib.lib().bbb_w_s2d_tap_major(w.data_ptr(), y.data_ptr(), E * Cout, Cin, kh, st, cur_stream(w.device)), "bbb_w_s2d_tap_major")
    return y


def c8x3_layer_ok(cin, cout, is_logits=False):
    """May a BBB layer with these channel counts run on bbb_conv2d_c8x3_fwd?  (16-channel k steps inside one tap; the output is
    written in groups of 8 channels unless it is the fp32 logits tensor.)"""
    return cin % 16 == 0 and (is_logits or cout % 8 == 0)


def conv2d_c8x3_forward(x, w_tm, bias, kernel_size, stride=1, padding=0, dilation=1, act=None, out_f32=False, out=None, units=None,
                        n_units=None, x_div=1, x_off=0, tile=None, nt=None, pool=False, x_per_slice=False, zero_border=(0, 0, 0, 0)):
    """The split-bf16 contraction over MFMA-ready operands (bbb_conv2d_c8x3_fwd).  x: c8 S3 [E|1, 3, Cin / 8, H, W, B, 8];
    w_tm: fp32 tap-major [E|1, Cout, kh * kw, Cin]; bias [E|1, Cout] | None -> c8 S3 [E, 3, Cout / 8, Ho, Wo, B, 8], or with
    out_f32 the fp32 batch-innermost [E, Cout, Ho, Wo, B] (the logits layer).  Work units / x_div / x_off as conv2d_chwn_forward.
    tile = 128 | 256: images per workgroup, nt = 2 | 3 | 4: 32-channel tiles per workgroup (None: the library picks by layer and
    launch size; same bits whatever the choice).
    pool = True (padding 0, even Ho and Wo, c8 S3 output): the launch also applies MaxPool2d(2, 2) to the activated output ->
    [E, 3, Cout / 8, Ho / 2, Wo / 2, B, 8], bit for bit maxpool_c8s3(conv2d_c8x3_forward(...), 2, 2); tile then means 32 | 64
    images per workgroup (its four waves own the four pixels of a window).
    zero_border = (leading rows, leading columns, trailing rows, trailing columns) of the input map that hold nothing but zeros (a
    space-to-depth block image's materialised padding, s2d_zero_border): their products are skipped -- same result, less work."""
    require_device(w_tm, bias)
    require_device(x, dtype=torch.bfloat16)
    x, w_tm = x.contiguous(), w_tm.contiguous()
    bias = None if bias is None else bias.contiguous()
    if x.dim() != 7 or x.shape[1] != 3 or x.shape[6] != 8 or w_tm.dim() != 4:
        raise _lib.BBBHipError("c8 S3 input [E|1, 3, Cin / 8, H, W, B, 8] and tap-major weights [E|1, Cout, kh * kw, Cin] expected")
    kh, kw = _pair(kernel_size)
    Ex, _, CG, H, W, B, _ = x.shape
    Ew, Cout, T, Cin = w_tm.shape
    if T != kh * kw or Cin != CG * 8:
        raise _lib.BBBHipError(f"weights [{Cout}, {T}, {Cin}] do not match a {kh} x {kw} layer on {CG * 8} channels")
    x5 = _Shape((Ex, Cin, H, W, B))
    w5 = _Shape((Ew, Cout, Cin, kh, kw))
    if units is not None and units[0] > 1:
        E = int(n_units)
        if Ex != (units[0] if x_per_slice else E):
            raise _lib.BBBHipError("work units: x must hold one slab per unit, or one per batch slice with x_per_slice")
        d, ho, wo = _desc_chwn(x5, w5, stride, padding, dilation, E, False, False, act)
        _apply_units(d, units, x_per_slice)
    elif int(x_div) > 1:
        E = Ew
        if not 0 <= int(x_off) < int(x_div) or Ex != -(-(E + int(x_off)) // int(x_div)):
            raise _lib.BBBHipError("x_div: x must hold ceil((E + x_off) / x_div) input slabs for the E weight sets")
        d, ho, wo = _desc_chwn(x5, w5, stride, padding, dilation, E, False, False, act)
        d.x_unit_div, d.x_unit_off = int(x_div), int(x_off)
    else:
        E = max(Ex, Ew)
        if Ex not in (1, E) or Ew not in (1, E):
            raise _lib.BBBHipError("leading (draw) dims of x and w must be 1 or equal")
        d, ho, wo = _desc_chwn(x5, w5, stride, padding, dilation, E, Ex == 1 and E > 1, Ew == 1 and E > 1, act)
    d.x_draw_stride *= 3                                       # bf16 elements per slab of three planes
    if pool:
        if out_f32 or ho % 2 or wo % 2 or _pair(padding) != (0, 0):
            raise _lib.BBBHipError("pool=True: c8 S3 output, no padding, even output height and width")
        ho, wo = ho // 2, wo // 2
    if out_f32:
        shape, odt = (E, Cout, ho, wo, B), torch.float32
    else:
        if Cout % 8:
            raise _lib.BBBHipError("a c8 S3 output needs a multiple of 8 output channels")
        shape, odt = (E, 3, Cout // 8, ho, wo, B, 8), torch.bfloat16
    if out is None:
        y = torch.empty(shape, dtype=odt, device=x.device)
    else:
        n = 1
        for v in shape:
            n *= v
        if out.numel() != n or not out.is_contiguous() or out.dtype != odt:
            raise _lib.BBBHipError("out= must be a contiguous tensor of the output's size and dtype")
        y = out.view(shape)
    with on_device(x.device):
        check(_lib.lib().bbb_conv2d_c8x3_fwd(ctypes.byref(d), x.data_ptr(), w_tm.data_ptr(), ptr(bias), y.data_ptr(),
                                             (1 if out_f32 else 0) | {None: 0, 128: 2, 256: 4, 32: 2, 64: 4}[tile] | (8 if pool else 0) |
                                             ({None: 0, 2: 2, 3: 3, 4: 4}[nt] << 4) | (min(15, zero_border[0]) << 8) |
                                             (min(15, zero_border[1]) << 12) | (min(15, zero_border[2]) << 16) | (min(15, zero_border[3]) << 20),
                                             cur_stream(x.device)),
              "bbb_conv2d_c8x3_fwd")
    return y


def maxpool_c8s3(x, k, s):
    """MaxPool2d(k, s) on a c8 S3 tensor [E, 3, C / 8, H, W, B, 8] (element-wise on the 16-byte channel vectors); six-plane slabs
    (the LRT chain): the maximum of the values, and the squares of the pooled values behind."""
    require_device(x, dtype=torch.bfloat16)
    x = x.contiguous()
    E, planes, CG, H, W, B, eight = x.shape
    if planes == 6:
        ho, wo = (H - k) // s + 1, (W - k) // s + 1
        y = torch.empty((E, 6, CG, ho, wo, B, 8), dtype=torch.bfloat16, device=x.device)
        with on_device(x.device):
            check(_lib.lib().bbb_maxpool_chwn_s3sq(x.data_ptr(), y.data_ptr(), E, CG, H, W, B * 8, int(k), int(s), cur_stream(x.device)),
                  "bbb_maxpool_chwn_s3sq")
        return y
    y = maxpool_chwn_s3(x.view(E, 3, CG, H, W, B * 8), k, s)
    return y.view(E, 3, CG, y.shape[3], y.shape[4], B, 8)


def lrt_conv2d_c8x3_forward(x, w_mu_tm, w_var_tm, b_mu, b_var, kernel_size, seed, call0, stream_id, stride=1, padding=0, dilation=1,
                            sample=True, act=None, out_f32=False, out=None, units=None, n_units=None, b_offset=0, x_per_slice=False,
                            x_div=1, x_off=0, n_slabs=None, tile=None, zero_border=(0, 0, 0, 0)):
    """The LRT layer on the split-bf16 kernel over MFMA-ready operands (bbb_lrt_conv2d_c8x3_fwd): both contractions -- (x, W_mu) and
    (x^2, W_sigma^2) -- on the 16-bit matrix pipe at fp32 accuracy, sampling epilogue as lrt_conv2d_chwn_forward's (same noise
    elements).  x: six-plane c8 S3 [E|1|S, 6, Cin / 8, H, W, B, 8]; w_mu_tm / w_var_tm: fp32 tap-major [Cout, kh * kw, Cin];
    -> six-plane c8 S3 [E, 6, Cout / 8, Ho, Wo, B, 8], or with out_f32 the fp32 [E, Cout, Ho, Wo, B].  Work units / x_div / x_off /
    b_offset / n_slabs as lrt_conv2d_chwn_forward."""
    require_device(w_mu_tm, w_var_tm, b_mu, b_var)
    require_device(x, dtype=torch.bfloat16)
    x, w_mu_tm, w_var_tm = x.contiguous(), w_mu_tm.contiguous(), w_var_tm.contiguous()
    b_mu = None if b_mu is None else b_mu.contiguous()
    b_var = None if b_var is None else b_var.contiguous()
    if x.dim() != 7 or x.shape[1] != 6 or x.shape[6] != 8 or w_mu_tm.dim() != 3 or w_var_tm.shape != w_mu_tm.shape:
        raise _lib.BBBHipError("six-plane c8 S3 input [E|1, 6, Cin / 8, H, W, B, 8] and tap-major weights [Cout, kh * kw, Cin] expected")
    kh, kw = _pair(kernel_size)
    Ex, _, CG, H, W, B, _ = x.shape
    Cout, T, Cin = w_mu_tm.shape
    if T != kh * kw or Cin != CG * 8:
        raise _lib.BBBHipError(f"weights [{Cout}, {T}, {Cin}] do not match a {kh} x {kw} layer on {CG * 8} channels")
    x5 = _Shape((Ex, Cin, H, W, B))
    w5 = _Shape((1, Cout, Cin, kh, kw))
    E = (int(n_slabs) if n_slabs is not None else Ex * int(x_div)) if (units is None or units[0] <= 1) else int(n_units)
    d, ho, wo = _desc_chwn(x5, w5, stride, padding, dilation, E, Ex == 1 and E > 1 and int(x_div) <= 1 and (units is None or units[0] <= 1),
                           True, act)
    if units is not None and units[0] > 1:
        if Ex != (units[0] if x_per_slice else E):
            raise _lib.BBBHipError("work units: x must hold one slab per unit, or one per batch slice with x_per_slice")
        _apply_units(d, units, x_per_slice)
    elif int(x_div) > 1:
        if not 0 <= int(x_off) < int(x_div) or Ex != -(-(E + int(x_off)) // int(x_div)):
            raise _lib.BBBHipError("x_div: x must hold ceil((E + x_off) / x_div) input slabs for the E output slabs")
        d.x_unit_div, d.x_unit_off = int(x_div), int(x_off)
    elif Ex not in (1, E):
        raise _lib.BBBHipError("x must hold one slab, or one per output slab")
    d.b_offset = int(b_offset)
    d.w_draw_stride = 0
    d.b_draw_stride = 0
    d.x_draw_stride *= 6                                       # bf16 elements per slab of six planes
    if out_f32:
        shape, odt = (E, Cout, ho, wo, B), torch.float32
    else:
        if Cout % 8:
            raise _lib.BBBHipError("a c8 S3 output needs a multiple of 8 output channels")
        shape, odt = (E, 6, Cout // 8, ho, wo, B, 8), torch.bfloat16
    if out is None:
        y = torch.empty(shape, dtype=odt, device=x.device)
    else:
        n = 1
        for v in shape:
            n *= v
        if out.numel() != n or not out.is_contiguous() or out.dtype != odt:
            raise _lib.BBBHipError("out= must be a contiguous tensor of the output's size and dtype")
        y = out.view(shape)
    flags = (1 if out_f32 else 0) | {None: 0, 128: 2, 256: 4}[tile] | (min(15, zero_border[0]) << 8) | (min(15, zero_border[1]) << 12) | \
        (min(15, zero_border[2]) << 16) | (min(15, zero_border[3]) << 20)
    with on_device(x.device):
        check(_lib.lib().bbb_lrt_conv2d_c8x3_fwd(ctypes.byref(d), x.data_ptr(), w_mu_tm.data_ptr(), w_var_tm.data_ptr(), ptr(b_mu), ptr(b_var),
                                                 y.data_ptr(), seed, call0 & 0xFFFFFFFF, stream_id, 1 if sample else 0,
                                                 rng.call_dev_ptr(x.device), flags, cur_stream(x.device)), "bbb_lrt_conv2d_c8x3_fwd")
    return y


def lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, seed, call0, stream_id, stride=1, padding=0, dilation=1,
                            sample=True, eps=None, want_moments=False, act=None, units=None, n_units=None, b_offset=0,
                            x_per_slice=False, x_div=1, x_off=0, n_slabs=None, pool=False):
    """LRT layer, batch-innermost.  x: [E, Cin, H, W, B] -> (y, act_mu|None, act_var|None) [E, Cout, Ho, Wo, B].
    Work units as in conv2d_chwn_forward (call0 = the call index of the first unit's draw; the noise of unit u is keyed by
    its draw and by the GLOBAL image index slice*B + b).  b_offset: global index of local image 0 (batch-parallel shards)."""
    require_device(x, w_mu, w_var, b_mu, b_var, eps)
    x = x.contiguous()
    w_mu, w_var = w_mu.contiguous(), w_var.contiguous()
    b_mu = None if b_mu is None else b_mu.contiguous()
    b_var = None if b_var is None else b_var.contiguous()
    E = (int(n_slabs) if n_slabs is not None else x.shape[0] * int(x_div)) if (units is None or units[0] <= 1) else int(n_units)
    d, ho, wo = _desc_chwn(x, w_mu.unsqueeze(0), stride, padding, dilation, E, False, True, act)
    if units is not None and units[0] > 1:
        _apply_units(d, units, x_per_slice)
    elif int(x_div) > 1:
        if not 0 <= int(x_off) < int(x_div) or x.shape[0] != -(-(E + int(x_off)) // int(x_div)):
            raise _lib.BBBHipError("x_div: x must hold ceil((E + x_off) / x_div) input slabs for the E output slabs")
        d.x_unit_div, d.x_unit_off = int(x_div), int(x_off)   # several steps per launch: slab e = step (e + x_off) // x_div on that step's batch
    d.b_offset = int(b_offset)
    d.w_draw_stride = 0
    d.b_draw_stride = 0
    if pool:
        # [activation ->] MaxPool2d(2, 2) inside the launch (pconv_body.cuh, POOL + LRT): the sampling epilogue per window pixel with
        # the CONV output's noise elements, running maximum in registers -- bit for bit maxpool_chwn of the unpooled launch
        if want_moments or ho % 2 or wo % 2:
            raise _lib.BBBHipError("pool=True: even output height / width, and no moment outputs (the unpooled pixels are not kept)")
        d.pool = 1
    shape = (E, w_mu.shape[0], ho // 2, wo // 2, x.shape[4]) if pool else (E, w_mu.shape[0], ho, wo, x.shape[4])
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    am = torch.empty(shape, dtype=torch.float32, device=x.device) if want_moments else None
    av = torch.empty(shape, dtype=torch.float32, device=x.device) if want_moments else None
    if eps is not None:
        eps = eps.contiguous()
    with on_device(x.device):
        ks, scr = _split_scratch(d, True, x.device)
        if pool and ks > 1:
            raise _lib.BBBHipError("pool=True: this layer's contraction is split (conv + maxpool_chwn instead)")
        check(_lib.lib().bbb_lrt_conv2d_chwn_splitk_fwd(ctypes.byref(d), x.data_ptr(), w_mu.data_ptr(), w_var.data_ptr(), ptr(b_mu),
                                                        ptr(b_var), y.data_ptr(), ptr(am), ptr(av), ptr(eps), seed,
                                                        call0 & 0xFFFFFFFF, stream_id, 1 if sample else 0,
                                                        rng.call_dev_ptr(x.device), ks, ptr(scr), 0 if scr is None else scr.numel(),
                                                        cur_stream(x.device)),
              "bbb_lrt_conv2d_chwn_splitk_fwd")
    return y, am, av


def lrt_sample_chwn(act_mu, act_var, draws, seed, call0, stream_id, act=None, b_offset=0):
    """E draws y[e] = act(act_mu + sqrt(act_var) * eps[e]) from one pair of LRT moments [1|-, C, Ho, Wo, B] ->
    [E, C, Ho, Wo, B]; eps as the LRT GEMM epilogue would draw it for draw e."""
    require_device(act_mu, act_var)
    act_mu, act_var = act_mu.contiguous(), act_var.contiguous()
    C, Ho, Wo, B = act_mu.shape[-4:]
    y = torch.empty((draws, C, Ho, Wo, B), dtype=torch.float32, device=act_mu.device)
    with on_device(act_mu.device):
        check(_lib.lib().bbb_lrt_sample_chwn(act_mu.data_ptr(), act_var.data_ptr(), y.data_ptr(), draws, C, Ho * Wo, B, int(b_offset),
                                             {None: 0, "relu": 1, "softplus": 2}[act], seed, call0 & 0xFFFFFFFF, stream_id,
                                             rng.call_dev_ptr(act_mu.device), cur_stream(act_mu.device)), "bbb_lrt_sample_chwn")
    return y


def maxpool_chwn(x, k, s):
    """MaxPool2d(k, s) on [..., H, W, B] (B innermost, B % 4 == 0)."""
    require_device(x)
    x = x.contiguous()
    *lead, H, W, B = x.shape
    planes = 1
    for v in lead:
        planes *= v
    ho, wo = (H - k) // s + 1, (W - k) // s + 1
    y = torch.empty((*lead, ho, wo, B), dtype=torch.float32, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_maxpool_chwn(x.data_ptr(), y.data_ptr(), planes, H, W, B, int(k), int(s), cur_stream(x.device)),
              "bbb_maxpool_chwn")
    return y


# ---- bf16 storage path (BASELINE.json configs[1]) -------------------------------------------------------------------

def bf16_row_pitch(k):
    return (int(k) + 7) & ~7


def bf16_tap_major(shape):
    """Conv weights [Cout, Cin, kh, kw] with Cin % 8 == 0 and more than one tap are stored tap-major ((r, q, ci) order
    inside a row): the bf16 GEMM can then skip the taps that fall into the padding."""
    return len(shape) == 4 and shape[1] % 8 == 0 and shape[2] * shape[3] > 1


def sample_weights_bf16(mus, rhos, prior_mu, prior_sigma, stream_ids, seed, call0, draws, eps=None):
    """The fused reparam + KL pass with the sampled WEIGHT matrices written as bf16 in the bf16 GEMM's operand layout
    ([draws, rows, pitch], pitch = row length rounded up to 8, pad zero; tap-major column order where bf16_tap_major says
    so) and 1-d tensors (biases) kept fp32.  Inference only.  Returns (kl, [tensors])."""
    if len(mus) == 0 or len(mus) > _lib.MAX_SEGMENTS:
        raise _lib.BBBHipError(f"1..{_lib.MAX_SEGMENTS} tensors per launch, got {len(mus)}")
    require_device(*mus, *rhos)
    dev = mus[0].device
    mus = [m.detach().contiguous() for m in mus]
    rhos = [r.detach().contiguous() for r in rhos]
    outs, row_lens, taps = [], [], []
    for m in mus:
        if m.dim() >= 2:
            rows, k = m.shape[0], m.numel() // m.shape[0]
            outs.append(torch.empty((draws, rows, bf16_row_pitch(k)), dtype=torch.bfloat16, device=dev))   # pad written by the kernel
            row_lens.append(k)
            taps.append(m.shape[2] * m.shape[3] if bf16_tap_major(tuple(m.shape)) else 0)
        else:
            outs.append(torch.empty((draws,) + tuple(m.shape), dtype=torch.float32, device=dev))
            row_lens.append(0)
            taps.append(0)
    if eps is not None:
        eps = [e.contiguous() for e in eps]
        require_device(*eps)
    segs = _segments(mus, rhos, outs, None, eps, stream_ids, draws)
    for i, (o, rl, tp) in enumerate(zip(outs, row_lens, taps)):
        if rl:
            segs[i].w_row_len = rl
            segs[i].w_taps = tp
            segs[i].draw_stride = o.shape[1] * o.shape[2]
    kl = torch.empty((), dtype=torch.float32, device=dev)
    L = _lib.lib()
    parts = _partials(dev, L.bbb_reparam_partials(segs, len(mus)))
    with on_device(dev):
        rc = L.bbb_reparam_kl_fwd(segs, len(mus), draws, float(prior_mu), float(prior_sigma), seed, call0 & 0xFFFFFFFF,
                                  0, ptr(parts), ptr(kl), 0, rng.call_dev_ptr(dev), cur_stream(dev))
    check(rc, "bbb_reparam_kl_fwd")
    return kl, outs


def sample_weights_tm(mus, rhos, prior_mu, prior_sigma, stream_ids, seed, call0, draws, tap_major):
    """The fused reparam + KL pass with the conv weights flagged in `tap_major` (one bool per tensor) written TAP-MAJOR, fp32
    [draws, Cout, kh * kw, Cin] (bbb_segment_t::w_tm_cin: the weight operand of conv2d_c8x3_forward), every other tensor dense
    [draws, *shape].  Same noise elements, same KL bits as the dense launch (w_tap_major(dense) == this, bit for bit).
    Inference only.  Returns (kl, [tensors])."""
    if len(mus) == 0 or len(mus) > _lib.MAX_SEGMENTS:
        raise _lib.BBBHipError(f"1..{_lib.MAX_SEGMENTS} tensors per launch, got {len(mus)}")
    require_device(*mus, *rhos)
    dev = mus[0].device
    mus = [m.detach().contiguous() for m in mus]
    rhos = [r.detach().contiguous() for r in rhos]
    outs = []
    for m, tm in zip(mus, tap_major):
        if tm:
            if m.dim() != 4 or m.shape[1] % 8 or m.shape[2] * m.shape[3] < 2:
                raise _lib.BBBHipError("tap-major output: conv weights [Cout, Cin, kh, kw] with Cin % 8 == 0 and more than one tap")
            outs.append(torch.empty((draws, m.shape[0], m.shape[2] * m.shape[3], m.shape[1]), dtype=torch.float32, device=dev))
        else:
            outs.append(torch.empty((draws,) + tuple(m.shape), dtype=torch.float32, device=dev))
    segs = _segments(mus, rhos, outs, None, None, stream_ids, draws)
    for i, (m, tm) in enumerate(zip(mus, tap_major)):
        if tm:
            segs[i].w_taps = m.shape[2] * m.shape[3]
            segs[i].w_tm_cin = m.shape[1]
    kl = torch.empty((), dtype=torch.float32, device=dev)
    L = _lib.lib()
    parts = _partials(dev, L.bbb_reparam_partials(segs, len(mus)))
    with on_device(dev):
        rc = L.bbb_reparam_kl_fwd(segs, len(mus), draws, float(prior_mu), float(prior_sigma), seed, call0 & 0xFFFFFFFF,
                                  0, ptr(parts), ptr(kl), 0, rng.call_dev_ptr(dev), cur_stream(dev))
    check(rc, "bbb_reparam_kl_fwd")
    return kl, outs


def to_batch_innermost_bf16(x):
    """fp32 [B, C, H, W] -> bf16 [C, H, W, B] (one rounding, nearest-even)."""
    require_device(x)
    x = x.contiguous()
    B = x.shape[0]
    plane = x.numel() // B
    y = torch.empty(tuple(x.shape[1:]) + (B,), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_nchw_to_chwn_bf16(x.data_ptr(), y.data_ptr(), B, plane, cur_stream(x.device)), "bbb_nchw_to_chwn_bf16")
    return y


def to_batch_innermost_bf16_slices(x, slices):
    """fp32 [S*Bs, C, H, W] -> bf16 [S, C, H, W, Bs] in one launch."""
    require_device(x)
    x = x.contiguous()
    Bs = x.shape[0] // slices
    plane = x.numel() // x.shape[0]
    y = torch.empty((slices,) + tuple(x.shape[1:]) + (Bs,), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_nchw_to_chwn_bf16_slices(x.data_ptr(), y.data_ptr(), Bs, plane, int(slices), cur_stream(x.device)),
              "bbb_nchw_to_chwn_bf16_slices")
    return y


def bf16_pool_fusion_ok(cin_khkw, tap_major, out_f32, pool_module, x_shape=None, geom=None, draws=1):
    """Should conv2d_chwn_bf16_forward(..., pool=(k, s)) replace conv + maxpool_chwn_bf16(k, s)?  First layers with a short
    contraction (row pitch <= 128: 3Conv3FC conv1, LeNet conv1 -- pconv_bf16_smallk_pool_kernel) followed by [activation ->]
    MaxPool2d(2, 2) or MaxPool2d(3, 2) without padding / dilation / ceil_mode.  Same values as the two launches (the maximum is
    taken before bias, activation and rounding, which are non-decreasing), so the choice never changes a result.
    x_shape [*, Cin, H, W, B] + geom (stride, padding, dilation) + draws: the launch (the thresholds of the current LaunchConfig;
    0 = always: the library picks the window-resident form for small launches and the strip form for large ones, and one of them
    beats the two launches at every size measured -- 3Conv3FC conv1 + pool1 at bs 256, us per launch: 1 step 18.4 -> 16.8, 4 steps
    46.5 -> 37.4, 16 steps 167 -> 125; profiles/r05_notes.md section 3).  Without x_shape: only whether the library HAS the form."""
    if not current_config().pool_fusion or pool_module is None or tap_major or out_f32:
        return False
    cin, kh, kw = cin_khkw
    if bf16_row_pitch(cin * kh * kw) > 128:
        return False
    pr = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    ks = (pr(pool_module.kernel_size), pr(pool_module.stride if pool_module.stride is not None else pool_module.kernel_size))
    if ks not in (((2, 2), (2, 2)), ((3, 3), (2, 2))):
        return False
    if not (pr(pool_module.padding) == (0, 0) and pr(pool_module.dilation) == (1, 1) and not pool_module.ceil_mode and
            not getattr(pool_module, "return_indices", False)):
        return False
    if x_shape is None:
        return True
    (sh, sw), (ph, pw), (dh, dw) = _pair(geom[0]), _pair(geom[1]), _pair(geom[2])
    H, W, B = x_shape[-3], x_shape[-2], x_shape[-1]
    ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    pk, pst = ks[0][0], ks[1][0]
    if ho < pk or wo < pk:
        return False
    rows = int(draws) * -(-B // 256) * ((ho - pk) // pst + 1)
    cfg = current_config()
    return rows >= (cfg.bf16_pool_fuse_min_rows_overlapped if cfg.launches_overlap else cfg.bf16_pool_fuse_min_rows)


def conv2d_chwn_bf16_forward(x, w, bias, cin_khkw, stride=1, padding=0, dilation=1, act=None, out_f32=False, out=None,
                             tap_major=False, units=None, n_units=None, x_per_slice=False, x_div=1, x_off=0, pool=None,
                             out_c8=False):
    """bf16 batch-innermost conv.  x: [E|1, Cin, H, W, B] bf16 (B % 8 == 0); w: [E|1, Cout, Kp] bf16 as written by
    sample_weights_bf16 (tap_major = its column order, see bf16_tap_major); cin_khkw = (Cin, kh, kw); bias [E|1, Cout]
    fp32 or None -> y [E, Cout, Ho, Wo, B] bf16 (fp32 when out_f32).
    pool = (k, s): the launch also applies MaxPool2d(k, s) to the activated output -> [E, Cout, Hp, Wp, B] (bf16_pool_fusion_ok
    says when the library has that form).
    Channel-interleaved activations ("c8", see to_c8): a 6-d x [E|1, Cin / 8, H, W, B, 8] is read as such (bf16_c8_input_ok says
    which layers have that form); out_c8=True writes y as [E, Cout / 8, Ho, Wo, B, 8] (bf16_c8_output_ok)."""
    require_device(x, w, dtype=torch.bfloat16)
    require_device(bias)
    x, w = x.contiguous(), w.contiguous()
    x_c8 = x.dim() == 6
    if x_c8:
        if x.shape[5] != 8:
            raise _lib.BBBHipError("a 6-d input is [E, Cin / 8, H, W, B, 8]")
        x = x.view(x.shape[0], x.shape[1] * 8, x.shape[2], x.shape[3], x.shape[4])      # same bytes; only the element order differs
    bias = None if bias is None else bias.contiguous()
    cin, kh, kw = cin_khkw
    sharded = units is not None and units[0] > 1
    grouped = not sharded and int(x_div) > 1
    E = int(n_units) if sharded else (w.shape[0] if grouped else max(x.shape[0], w.shape[0]))
    if grouped and (not 0 <= int(x_off) < int(x_div) or x.shape[0] != -(-(E + int(x_off)) // int(x_div))):
        raise _lib.BBBHipError("x_div: x must hold ceil((E + x_off) / x_div) input slabs for the E weight sets")
    if not sharded and not grouped and (x.shape[0] not in (1, E) or w.shape[0] not in (1, E)):
        raise _lib.BBBHipError("leading (draw) dims of x and w must be 1 or equal")
    Ex, Cin, H, W, B = x.shape
    if Cin != cin or w.shape[2] != bf16_row_pitch(cin * kh * kw):
        raise _lib.BBBHipError("weight pitch / channel count do not match the geometry")
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    d = ConvDesc()
    d.batch, d.cin, d.h, d.w, d.cout, d.kh, d.kw = B, Cin, H, W, w.shape[1], kh, kw
    d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.dil_h, d.dil_w = sh, sw, ph, pw, dh, dw
    d.draws = E
    d.x_draw_stride = 0 if (Ex == 1 and E > 1 and not sharded and not grouped) else Cin * H * W * B
    if grouped:
        d.x_unit_div, d.x_unit_off = int(x_div), int(x_off)
    d.w_draw_stride = 0 if (w.shape[0] == 1 and E > 1 and not sharded) else w.shape[1] * w.shape[2]
    d.b_draw_stride = 0 if (bias is None or (bias.shape[0] == 1 and E > 1 and not sharded)) else w.shape[1]
    d.act = {None: 0, "relu": 1, "softplus": 2}[act]
    if sharded:
        _apply_units(d, units, x_per_slice)
    ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    if pool is not None:
        pk, pst = int(pool[0]), int(pool[1])
        d.pool = 1 if (pk, pst) == (2, 2) else ((pk << 8) | pst)
        ho, wo = (ho - pk) // pst + 1, (wo - pk) // pst + 1
    if out_c8 and (out_f32 or w.shape[1] % 8):
        raise _lib.BBBHipError("out_c8: bf16 output with a multiple of 8 channels")
    shape = (E, w.shape[1] // 8, ho, wo, B, 8) if out_c8 else (E, w.shape[1], ho, wo, B)
    dt = torch.float32 if out_f32 else torch.bfloat16
    if out is None:
        y = torch.empty(shape, dtype=dt, device=x.device)
    else:
        if out.numel() != E * w.shape[1] * ho * wo * B or not out.is_contiguous() or out.dtype != dt:
            raise _lib.BBBHipError("out= must be a contiguous tensor of the output's size and dtype")
        y = out.view(shape)
    with on_device(x.device):
        check(_lib.lib().bbb_conv2d_chwn_bf16_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), ptr(bias), y.data_ptr(),
                                                  (1 if out_f32 else 0) | (2 if tap_major else 0) | (4 if x_c8 else 0) |
                                                  (8 if out_c8 else 0), cur_stream(x.device)),
              "bbb_conv2d_chwn_bf16_fwd")
    return y


def bf16_c8_input_ok(cin_khkw, geom, tap_major, out_f32, x_shape=None, draws=1):
    """Does conv2d_chwn_bf16_forward have a kernel that reads THIS layer's input channel-interleaved (a 6-d x, see to_c8), and
    should the launch take it?  pconv_bf16_strip8_kernel: tap-major rows of 32 input channels, 5 x 5 taps, stride 1, no dilation,
    padding < 5, bf16 output (3Conv3FC conv2: 131-153 us on the general kernel at bs 256 x 16 steps per launch, 88-100 us here).  The
    producer must then write that layout (out_c8: the pooled first-layer forms and the strip form itself).
    x_shape (H, W, B) + draws: the launch, for LaunchConfig.bf16_c8_min_items (strip workgroups: a strip of three pixels of an output
    row x 64 channels x 128 images).  0 = always: 3Conv3FC conv2 at bs 256, us per launch, general kernel / strip form: 1 step 18.1 /
    15.0, 2 steps 33.7 / 20.8, 4 steps 43.4 / 30.9, 8 steps 71.3 / 47.8, 16 steps 131-153 / 88-100 (profiles/r05_notes.md section 8).
    Values: the strip form always adds a pixel's taps in one fixed order (the general kernel's order with ONE k-group, which is what
    that kernel uses for launches of 512 workgroups and more) -- bit-identical there, rounding-level differences against the
    general kernel's two- / four-group launches."""
    cfg = current_config()
    if not cfg.bf16_c8 or not tap_major or out_f32:
        return False
    cin, kh, kw = cin_khkw
    (sh, sw), (ph, pw), (dh, dw) = _pair(geom[0]), _pair(geom[1]), _pair(geom[2])
    if not (cin == 32 and (kh, kw) == (5, 5) and (sh, sw) == (1, 1) and (dh, dw) == (1, 1) and ph < kh and pw < kw):
        return False
    if x_shape is None:
        return True
    H, W, B = x_shape
    ho, wo = H + 2 * ph - kh + 1, W + 2 * pw - kw + 1
    if ho < 1 or wo < 1:
        return False
    return int(draws) * ho * -(-wo // 3) * -(-B // 128) >= cfg.bf16_c8_min_items


def to_c8(x):
    """[E, C, H, W, B] -> the channel-interleaved layout [E, C / 8, H, W, B, 8] (8 consecutive channels of an image adjacent in
    memory: an MFMA operand of the bf16 kernels is then one 16-byte load).  A torch permute: tests and one-off conversions; inside
    the batched path the producing kernel writes the layout itself (out_c8)."""
    E, C, H, W, B = x.shape
    return x.view(E, C // 8, 8, H, W, B).permute(0, 1, 3, 4, 5, 2).contiguous()


def from_c8(x):
    """The inverse of to_c8: [E, C / 8, H, W, B, 8] -> [E, C, H, W, B]."""
    E, C8, H, W, B, _ = x.shape
    return x.permute(0, 1, 5, 2, 3, 4).reshape(E, C8 * 8, H, W, B).contiguous()


def maxpool_chwn_bf16(x, k, s):
    """MaxPool2d(k, s) on bf16 [..., H, W, B] (B % 8 == 0)."""
    require_device(x, dtype=torch.bfloat16)
    x = x.contiguous()
    *lead, H, W, B = x.shape
    planes = 1
    for v in lead:
        planes *= v
    ho, wo = (H - k) // s + 1, (W - k) // s + 1
    y = torch.empty((*lead, ho, wo, B), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_maxpool_chwn_bf16(x.data_ptr(), y.data_ptr(), planes, H, W, B, int(k), int(s), cur_stream(x.device)),
              "bbb_maxpool_chwn_bf16")
    return y


def mc_tail_cb(logits, mean_over=0, step_end=None):
    """mc_tail for batch-innermost logits [E, C, B] -> [B, C]."""
    return mc_tail_units(logits, 1, 0, mean_over, step_end)


class _McTailCB(torch.autograd.Function):
    """lse = mc_tail_cb(logits [E, C, B]) with its backward in ONE launch (bbb_mc_tail_cb_bwd): the training step's loss tail
    (log_softmax + logmeanexp, main_bayesian.py:49-53) was ~15 ATen launches forward and backward."""

    @staticmethod
    def forward(ctx, logits, mean_over):
        logits = logits.contiguous()
        lse = mc_tail_cb(logits, mean_over=mean_over)
        ctx.save_for_backward(logits, lse)
        ctx.mean_over = int(mean_over)
        return lse

    @staticmethod
    def backward(ctx, g):
        logits, lse = ctx.saved_tensors
        g = g.contiguous()
        E, C, B = logits.shape
        out = torch.empty_like(logits)
        with on_device(logits.device):
            check(_lib.lib().bbb_mc_tail_cb_bwd(logits.data_ptr(), lse.data_ptr(), g.data_ptr(), out.data_ptr(), E, B, C, ctx.mean_over,
                                                cur_stream(logits.device)), "bbb_mc_tail_cb_bwd")
        return out, None


class _ElboCB(torch.autograd.Function):
    """(loss, lse) = the whole loss tail of a training step: log_softmax + logmeanexp (bbb_mc_tail_cb), then nll_loss(mean) *
    train_size + beta * kl in one block (bbb_elbo_cb_fwd); backward in ONE launch (bbb_elbo_cb_bwd: d loss / d logits and
    d loss / d kl).  main_bayesian.py:49-56 / metrics.py:7-14 upstream -- ~25 ATen launches forward and backward otherwise."""

    @staticmethod
    def forward(ctx, logits, kl, target, beta, train_size, mean_over):
        logits = logits.contiguous()
        E, C, B = logits.shape
        lse = mc_tail_cb(logits, mean_over=mean_over)
        klc = kl.detach().reshape(()).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        beta_t = beta if torch.is_tensor(beta) else None
        with on_device(logits.device):
            check(_lib.lib().bbb_elbo_cb_fwd(lse.data_ptr(), target.data_ptr(), klc.data_ptr(), 0.0 if beta_t is not None else float(beta),
                                             ptr(beta_t), float(train_size), B, C, loss.data_ptr(), cur_stream(logits.device)),
                  "bbb_elbo_cb_fwd")
        ctx.save_for_backward(logits, lse, target)
        ctx.beta, ctx.beta_t, ctx.train_size, ctx.mean_over = (None if beta_t is not None else float(beta)), beta_t, float(train_size), int(mean_over)
        ctx.kl_shape = tuple(kl.shape)
        ctx.mark_non_differentiable(lse)
        return loss, lse

    @staticmethod
    def backward(ctx, g_loss, _g_lse):
        logits, lse, target = ctx.saved_tensors
        E, C, B = logits.shape
        g_loss = g_loss.to(dtype=torch.float32).reshape(()).contiguous()
        out = torch.empty_like(logits)
        g_kl = torch.empty(ctx.kl_shape, dtype=torch.float32, device=logits.device)
        with on_device(logits.device):
            check(_lib.lib().bbb_elbo_cb_bwd(logits.data_ptr(), lse.data_ptr(), target.data_ptr(), g_loss.data_ptr(),
                                             0.0 if ctx.beta_t is not None else ctx.beta, ptr(ctx.beta_t), ctx.train_size, out.data_ptr(),
                                             g_kl.data_ptr(), E, B, C, ctx.mean_over, cur_stream(logits.device)), "bbb_elbo_cb_bwd")
        return out, g_kl, None, None, None, None


def elbo_cb_ok(logits, kl, target, beta):
    """The fused loss tail takes: logits [E <= 512, C, B] fp32 on the device, a one-element fp32 kl there too, int64 targets [B] there too,
    beta a Python number or a one-element fp32 device tensor."""
    return (torch.is_tensor(logits) and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 3 and logits.shape[0] <= 512
            and torch.is_tensor(kl) and kl.device == logits.device and kl.dtype == torch.float32 and kl.numel() == 1
            and torch.is_tensor(target) and target.device == logits.device and target.dtype == torch.int64
            and target.dim() == 1 and target.shape[0] == logits.shape[2] and target.is_contiguous()
            and (not torch.is_tensor(beta) or (beta.device == logits.device and beta.dtype == torch.float32 and beta.numel() == 1
                                               and not beta.requires_grad)))


def elbo_cb_autograd(logits, kl, target, beta, train_size, mean_over=0):
    """Differentiable (loss, lse [B, C]) of one Monte-Carlo training step from its logits [E, C, B] and the KL of one forward."""
    require_device(logits)
    if not elbo_cb_ok(logits, kl, target, beta):
        raise _lib.BBBHipError("elbo_cb_autograd: see elbo_cb_ok for the accepted operands")
    return _ElboCB.apply(logits, kl, target, beta, float(train_size), int(mean_over))


def mc_tail_cb_autograd(logits, mean_over=0):
    """Differentiable mc_tail_cb: logits [E, C, B] (requires grad) -> lse [B, C]."""
    require_device(logits)
    if logits.shape[0] > 512:
        raise _lib.BBBHipError("too many draws for one tail launch")
    return _McTailCB.apply(logits, int(mean_over))


def mc_tail_units(logits, slices, unit_off, mean_over=0, step_end=None):
    """mc_tail_cb for a rank's work units: logits [U, C, Bs] (unit u = unit_off + e is draw u // slices, batch slice
    u % slices) -> [slices * Bs, C]; -inf rows for slices the rank holds no unit of.
    step_end = (kl_one_forward, scale, counter | None, counter_add): the end of a captured Monte-Carlo step in the same launch
    (bbb_mc_tail_units_step) -> returns (lse, kl_one_forward * scale) and adds counter_add to the int32 device counter."""
    require_device(logits)
    logits = logits.contiguous()
    U, C, Bs = logits.shape
    if U > (512 if int(slices) == 1 else 4096):
        raise _lib.BBBHipError("too many draws / units for one tail launch")
    out = torch.empty((int(slices) * Bs, C), dtype=torch.float32, device=logits.device)
    L = _lib.lib()
    with on_device(logits.device):
        if step_end is None:
            check(L.bbb_mc_tail_units(logits.data_ptr(), U, int(slices), int(unit_off) % int(slices), Bs, C, int(mean_over),
                                      out.data_ptr(), cur_stream(logits.device)), "bbb_mc_tail_units")
            return out
        kl_in, scale, counter, add = step_end
        require_device(kl_in)
        kl_out = torch.empty((), dtype=torch.float32, device=logits.device)
        if counter is not None and (counter.dtype != torch.int32 or not counter.is_cuda):
            raise _lib.BBBHipError("the call counter must be an int32 device tensor")
        check(L.bbb_mc_tail_units_step(logits.data_ptr(), U, int(slices), int(unit_off) % int(slices), Bs, C, int(mean_over),
                                       out.data_ptr(), kl_in.data_ptr(), float(scale), kl_out.data_ptr(), ptr(counter),
                                       int(add) & 0xFFFFFFFF, cur_stream(logits.device)), "bbb_mc_tail_units_step")
    return out, kl_out


def mc_tail_groups(logits, groups, draws, mean_over=0, step_end=None):
    """Tail of `groups` Monte-Carlo steps that share one set of launches: logits [groups * draws, C, B] (slab g * draws + j =
    draw j of step g) -> [groups * B, C], block g = log-sum-exp over step g's draws of the per-draw log_softmax, minus
    log(mean_over) when > 0.  step_end as in mc_tail_units."""
    require_device(logits)
    logits = logits.contiguous()
    U, C, B = logits.shape
    if U != int(groups) * int(draws) or U > 4096:
        raise _lib.BBBHipError("mc_tail_groups: logits must hold groups * draws <= 4096 slabs")
    out = torch.empty((int(groups) * B, C), dtype=torch.float32, device=logits.device)
    kl_in, scale, counter, add = step_end if step_end is not None else (None, 0.0, None, 0)
    require_device(kl_in)
    kl_out = torch.empty((), dtype=torch.float32, device=logits.device) if step_end is not None else None
    if counter is not None and (counter.dtype != torch.int32 or not counter.is_cuda):
        raise _lib.BBBHipError("the call counter must be an int32 device tensor")
    with on_device(logits.device):
        check(_lib.lib().bbb_mc_tail_groups_step(logits.data_ptr(), int(groups), int(draws), B, C, int(mean_over), out.data_ptr(),
                                                 ptr(kl_in), float(scale), ptr(kl_out), ptr(counter), int(add) & 0xFFFFFFFF,
                                                 cur_stream(logits.device)), "bbb_mc_tail_groups_step")
    return out if step_end is None else (out, kl_out)


def mc_tail_share(logits, steps, draws, first_off, step_end=None):
    """Tail of one rank's share of a GROUP of steps: logits [slabs, C, B], slab e = draw (first_off + e) of the draw-major (step,
    draw) enumeration of `steps` local steps with `draws` draws each -> [steps * B, C], block k = log-sum-exp over the LOCAL draws
    of local step k of the per-draw log_softmax (no mean: the ranks' blocks are combined by one more log-sum-exp; -inf where the
    share holds no draw of the step).  step_end as in mc_tail_units."""
    require_device(logits)
    logits = logits.contiguous()
    U, C, B = logits.shape
    if U > 4096 or not 0 <= int(first_off) < int(draws) or int(steps) * int(draws) < U + int(first_off):
        raise _lib.BBBHipError("mc_tail_share: the slabs must fit the steps * draws grid, <= 4096 slabs")
    out = torch.empty((int(steps) * B, C), dtype=torch.float32, device=logits.device)
    kl_in, scale, counter, add = step_end if step_end is not None else (None, 0.0, None, 0)
    require_device(kl_in)
    kl_out = torch.empty((), dtype=torch.float32, device=logits.device) if step_end is not None else None
    if counter is not None and (counter.dtype != torch.int32 or not counter.is_cuda):
        raise _lib.BBBHipError("the call counter must be an int32 device tensor")
    with on_device(logits.device):
        check(_lib.lib().bbb_mc_tail_share_step(logits.data_ptr(), U, int(steps), int(draws), int(first_off), B, C, out.data_ptr(),
                                                ptr(kl_in), float(scale), ptr(kl_out), ptr(counter), int(add) & 0xFFFFFFFF,
                                                cur_stream(logits.device)), "bbb_mc_tail_share_step")
    return out if step_end is None else (out, kl_out)


def uncertainty(logits, normalized=False):
    """logits [T, B, C] -> (pred, epistemic, aleatoric), each [B, C] (see bbb_uncertainty)."""
    require_device(logits)
    logits = logits.contiguous()
    T, B, C = logits.shape
    outs = [torch.empty((B, C), dtype=torch.float32, device=logits.device) for _ in range(3)]
    with on_device(logits.device):
        check(_lib.lib().bbb_uncertainty(logits.data_ptr(), T, B, C, 1 if normalized else 0, outs[0].data_ptr(),
                                         outs[1].data_ptr(), outs[2].data_ptr(), cur_stream(logits.device)), "bbb_uncertainty")
    return tuple(outs)


def to_batch_innermost(x):
    """[B, C, H, W] -> [C, H, W, B] (LDS-tiled transpose)."""
    require_device(x)
    x = x.contiguous()
    B = x.shape[0]
    out = torch.empty(tuple(x.shape[1:]) + (B,), dtype=torch.float32, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_transpose2d(x.data_ptr(), out.data_ptr(), B, x.numel() // B, cur_stream(x.device)), "bbb_transpose2d")
    return out


def from_batch_innermost(y):
    """[C, H, W, B] -> [B, C, H, W] (the same LDS-tiled transpose, the other way round)."""
    require_device(y)
    y = y.contiguous()
    B = y.shape[-1]
    out = torch.empty((B,) + tuple(y.shape[:-1]), dtype=torch.float32, device=y.device)
    with on_device(y.device):
        check(_lib.lib().bbb_transpose2d(y.data_ptr(), out.data_ptr(), y.numel() // B, B, cur_stream(y.device)), "bbb_transpose2d")
    return out


def _layer_on_fast_kernels(x4, *operands):
    """The per-layer path of the drop-in layers (a model with forward hooks, modules the whole-model path does not know, ...):
    may THIS layer's convolution run on the batch-innermost kernel between two layout transposes?  Inference only (the
    reference-layout kernels carry the autograd wiring), a 4-d fp32 GPU batch with B % 4 == 0."""
    if not (torch.is_tensor(x4) and x4.is_cuda and x4.dim() == 4 and x4.dtype == torch.float32 and x4.shape[0] % 4 == 0 and x4.shape[0] > 0):
        return False
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x4,) + operands):
        return False
    return True


def conv2d_layer(x4, w, b, stride, padding, dilation):
    """BBB conv layer of the per-layer path: x4 [B, Cin, H, W], w [1, Cout, Cin, kh, kw], b [1, Cout] | None -> [B, Cout, Ho, Wo].
    Inference takes the pixel-major batch-innermost kernel (padding taps skipped, the layer's split contraction: the kernel of
    the whole-model path) between two transposes -- 3-5x faster than the reference-layout kernel at bs 512; everything else the
    reference-layout autograd function."""
    if _layer_on_fast_kernels(x4, w, b):
        # (the plain fmaf chain, not the layer's split contraction: bit for bit the reference-layout kernel's result, which is what
        # this path promises -- tests/test_gpu_models.py::test_batched_ensemble_equals_loop)
        with use_config(split_k=False):
            y = conv2d_chwn_forward(to_batch_innermost(x4).unsqueeze(0), w, b, stride, padding, dilation, bf16x3=False)
        return from_batch_innermost(y[0])
    return conv2d(x4.unsqueeze(0), w, b, stride, padding, dilation).squeeze(0)


def lrt_conv2d_layer(x4, w_mu, w_var, b_mu, b_var, seed, call, stream_id, stride, padding, dilation, sample, eps):
    """LRT conv layer of the per-layer path (same noise elements on either kernel: the canonical NCHW index keys them)."""
    if eps is None and _layer_on_fast_kernels(x4, w_mu, w_var, b_mu, b_var):
        with use_config(split_k=False):
            y = lrt_conv2d_chwn_forward(to_batch_innermost(x4).unsqueeze(0), w_mu, w_var, b_mu, b_var, seed, call, stream_id, stride,
                                        padding, dilation, sample=sample)[0]
        return from_batch_innermost(y[0])
    return lrt_conv2d(x4.unsqueeze(0), w_mu, w_var, b_mu, b_var, seed, call, stream_id, stride, padding, dilation, sample=sample,
                      eps=eps).squeeze(0)


def to_batch_innermost_slices(x, slices):
    """[S*Bs, C, H, W] -> [S, C, H, W, Bs]: one batch-innermost block per batch slice (the input of a rank's work units), as ONE
    batched transpose launch."""
    require_device(x)
    x = x.contiguous()
    Bs = x.shape[0] // slices
    plane = x.numel() // x.shape[0]
    out = torch.empty((slices,) + tuple(x.shape[1:]) + (Bs,), dtype=torch.float32, device=x.device)
    if not _transpose_batched(x, out, Bs, plane, slices, 1, Bs * plane, 0, plane, plane * Bs, 0, Bs):
        raise _lib.BBBHipError("to_batch_innermost_slices: too many slices")
    return out


def mc_tail(logits, mean_over=0):
    """logits [E, B, C] -> [B, C]: log-sum-exp over draws of the per-draw log_softmax, minus log(mean_over)
    when mean_over > 0 (== utils.logmeanexp of the stacked log_softmax when mean_over == E)."""
    require_device(logits)
    logits = logits.contiguous()
    E, B, C = logits.shape
    out = torch.empty((B, C), dtype=torch.float32, device=logits.device)
    with on_device(logits.device):
        check(_lib.lib().bbb_mc_tail(logits.data_ptr(), E, B, C, int(mean_over), out.data_ptr(), cur_stream(logits.device)),
              "bbb_mc_tail")
    return out


# ------------------------------------------------------------------------------------------------
# autograd wiring
# ------------------------------------------------------------------------------------------------
class _SampleWeights(torch.autograd.Function):
    """(mu_0, rho_0, mu_1, rho_1, ...) -> (kl, w_0, w_1, ...), one fused launch; w_i: [draws, *shape]."""

    @staticmethod
    def forward(ctx, cfg, *params):
        mus, rhos = list(params[0::2]), list(params[1::2])
        ws, _, kl = reparam_kl_forward(mus, rhos, cfg["prior_mu"], cfg["prior_sigma"], cfg["stream_ids"], cfg["seed"],
                                       cfg["call0"], cfg["draws"], sample=True, eps=cfg.get("eps"),
                                       textbook_kl=cfg.get("textbook_kl", False))
        ctx.cfg = cfg
        ctx.save_for_backward(*params)
        return (kl, *ws)

    @staticmethod
    def backward(ctx, gkl, *gws):
        params = ctx.saved_tensors
        cfg = ctx.cfg
        mus, rhos = list(params[0::2]), list(params[1::2])
        gmu, grho = reparam_kl_backward(mus, rhos, list(gws), gkl, cfg["prior_mu"], cfg["prior_sigma"], cfg["stream_ids"],
                                        cfg["seed"], cfg["call0"], cfg["draws"], eps=cfg.get("eps"),
                                        textbook_kl=cfg.get("textbook_kl", False))
        out = [None]
        for a, b in zip(gmu, grho):
            out += [a, b]
        return tuple(out)


class _KLOnly(torch.autograd.Function):
    """KL (and optionally sigma / sigma^2 as a differentiable output) without sampling."""

    @staticmethod
    def forward(ctx, cfg, *params):
        mus, rhos = list(params[0::2]), list(params[1::2])
        _, sig, kl = reparam_kl_forward(mus, rhos, cfg["prior_mu"], cfg["prior_sigma"], cfg["stream_ids"], 0, 0, 1,
                                        sample=False, want_sigma=cfg.get("want_sigma", False),
                                        sigma_squared=cfg.get("sigma_squared", False),
                                        textbook_kl=cfg.get("textbook_kl", False))
        ctx.cfg = cfg
        ctx.save_for_backward(*params)
        ctx.n_sig = 0 if sig is None else len(sig)
        out = (kl,) if sig is None else (kl, *sig)
        if cfg.get("mu_through", False):
            # the means again, as outputs of THIS node: what consumes them (the LRT forward node) hands its gradients back here,
            # where the backward kernel adds them to d KL / d mu -- instead of one accumulation launch per tensor in the engine
            out = out + tuple(m.view_as(m) for m in mus)
        return out

    @staticmethod
    def backward(ctx, gkl, *gout):
        params = ctx.saved_tensors
        cfg = ctx.cfg
        mus, rhos = list(params[0::2]), list(params[1::2])
        gsig, gthru = gout[:ctx.n_sig], gout[ctx.n_sig:]
        # d sigma / d rho = sigmoid(rho), d sigma^2 / d rho = 2 sigma sigmoid(rho): folded into the backward kernel (the LRT
        # training step spent ~70 ATen launches per step on these twelve small tensors)
        gs = None
        if gsig and any(g is not None for g in gsig):
            gs = [gsig[i] if i < len(gsig) else None for i in range(len(mus))]
        gws = [None] * len(mus)
        if gthru and any(g is not None for g in gthru):
            gws = [gthru[i] if i < len(gthru) else None for i in range(len(mus))]
        gmu, grho = reparam_kl_backward(mus, rhos, gws, gkl, cfg["prior_mu"], cfg["prior_sigma"],
                                        cfg["stream_ids"], 0, 0, 1, textbook_kl=cfg.get("textbook_kl", False), gsigmas=gs,
                                        sigma_squared=cfg.get("sigma_squared", False), gws_mean_only=True)
        out = [None]
        for a, b in zip(gmu, grho):
            out += [a, b]
        return tuple(out)


def conv2d_splitk(x, w, bias, stride=1, padding=0, dilation=1):
    """conv2d_forward for the training path, where launches are small (one draw, a few hundred images): when the GEMM
    would occupy fewer than ~512 workgroups, the input channels are split into S chunks that run as extra "draws" of one
    launch and are summed afterwards in a fixed order (deterministic).  Same result up to fp32 summation order -- which is
    why the inference paths, whose loop / batched forms must agree bit for bit, never come through here."""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ex, B, Cin, H, W = x.shape
    Ew, Cout, _, kh, kw = w.shape
    E = max(Ex, Ew)
    ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    tiles = E * -(-B * ho * wo // 64) * -(-Cout // 64)
    S = 1
    while tiles * S < 512 and Cin % (2 * S) == 0 and (Cin // (2 * S)) * kh * kw >= 64:
        S *= 2
    if S == 1:
        return conv2d_forward(x, w, bias, stride, padding, dilation)
    c = Cin // S
    xs = x.reshape(Ex, B, S, c, H, W).permute(0, 2, 1, 3, 4, 5)                      # [Ex, S, B, c, H, W]
    ws = w.reshape(Ew, Cout, S, c, kh, kw).permute(0, 2, 1, 3, 4, 5)                 # [Ew, S, Cout, c, kh, kw]
    xs = xs.expand(E, S, B, c, H, W).reshape(E * S, B, c, H, W)
    ws = ws.expand(E, S, Cout, c, kh, kw).reshape(E * S, Cout, c, kh, kw)
    y = conv2d_forward(xs, ws, None, stride, padding, dilation)                      # [E*S, B, Cout, ho, wo]
    y = y.reshape(E, S, B, Cout, ho, wo).sum(1)
    if bias is not None:
        y = y + bias.reshape(bias.shape[0], 1, Cout, 1, 1)
    return y


def _small_launch(x, w, stride, padding, dilation):
    """True when a conv of x [E,B,Cin,H,W] with w [*,Cout,Cin,kh,kw] would occupy fewer than 512 64x64 output tiles and its
    channels can be split (the conv2d_splitk criterion)."""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    E, B, Cin, H, W = x.shape
    Cout, kh, kw = w.shape[1], w.shape[3], w.shape[4]
    ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    tiles = E * -(-B * ho * wo // 64) * -(-Cout // 64)
    return tiles < 512 and Cin % 2 == 0 and (Cin // 2) * kh * kw >= 64


def lrt_sample_nchw(act_mu, act_var, seed, call0, stream_id):
    """y = act_mu + sqrt(act_var) * eps for [E, ...] moments; eps = the LRT kernels' noise stream (draw e = call0 + e)."""
    require_device(act_mu, act_var)
    act_mu, act_var = act_mu.contiguous(), act_var.contiguous()
    E = act_mu.shape[0]
    n = act_mu.numel() // E
    y = torch.empty_like(act_mu)
    with on_device(act_mu.device):
        check(_lib.lib().bbb_lrt_sample_nchw(act_mu.data_ptr(), act_var.data_ptr(), y.data_ptr(), n, E, seed, call0 & 0xFFFFFFFF,
                                             stream_id, rng.call_dev_ptr(act_mu.device), cur_stream(act_mu.device)),
              "bbb_lrt_sample_nchw")
    return y


def conv2d_input_grad(gy, w, x_shape, stride, padding, dilation):
    """d loss / d x of y = conv2d(x, w) on the same fp32-MFMA kernel: a stride-1 convolution of gy -- zero-upsampled by the
    layer's stride -- with the spatially flipped, channel-transposed weights, the layer's dilation and padding
    d*(k-1) - p; rows / columns of x that no output touched get zero.  gy [E,B,Cout,Ho,Wo], w [E|1,Cout,Cin,kh,kw].
    (Stride s multiplies s*s - 1 inserted zeros: fine for the rare strided layer that is not a model's first layer -- a
    first layer needs no input gradient at all.)"""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    kh, kw = w.shape[3], w.shape[4]
    H, W = x_shape[-2], x_shape[-1]
    E, B, Cout, Ho, Wo = gy.shape
    # upsampled extent, extended so that the stride-1 transposed convolution yields exactly H x W: input rows past
    # (Ho-1)*s + d*(k-1) - p still receive gradient from the last outputs only if they exist, i.e. zeros are appended
    Uh, Uw = H + 2 * ph - dh * (kh - 1), W + 2 * pw - dw * (kw - 1)
    if (sh, sw) != (1, 1) or (Uh, Uw) != (Ho, Wo):
        up = gy.new_zeros((E, B, Cout, Uh, Uw))
        up[:, :, :, :(Ho - 1) * sh + 1:sh, :(Wo - 1) * sw + 1:sw] = gy
        gy = up
    qh, qw = dh * (kh - 1) - ph, dw * (kw - 1) - pw                      # padding of the transposed convolution
    w_t = w.flip(3, 4).transpose(1, 2).contiguous()                     # [E|1, Cin, Cout, kh, kw]
    gx = conv2d_splitk(gy, w_t, None, 1, (max(qh, 0), max(qw, 0)), (dh, dw))
    if qh < 0 or qw < 0:                                                 # padding larger than the kernel reach: crop
        gx = gx[:, :, :, max(-qh, 0):gx.shape[3] - max(-qh, 0), max(-qw, 0):gx.shape[4] - max(-qw, 0)]
    if gx.shape[3] != H or gx.shape[4] != W:
        raise _lib.BBBHipError("conv2d_input_grad: geometry mismatch")
    return gx.contiguous()


def conv2d_weight_grad(gy, x, w_shape, stride, padding, dilation):
    """d loss / d w on the same kernel: with batch and channel axes swapped, wgrad is itself a convolution --
    input x^T [Cin as batch, B as channels, H, W], "weights" gy^T [Cout, B, Ho, Wo], stride = the layer's dilation,
    dilation = the layer's stride -> [Cin, Cout, kh(+), kw(+)], cropped and transposed back.  One GEMM per draw, batched
    over draws; when the weights are shared by all draws (LRT) the draw axis is folded into the batch."""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ew, Cout, Cin, kh, kw = w_shape
    E = gy.shape[0]
    if x.shape[0] == 1 and E > 1:
        x = x.expand(E, *x.shape[1:])
    if Ew == 1 and E > 1:                                                # shared weights: sum over draws = bigger batch
        x = x.reshape(1, E * x.shape[1], *x.shape[2:])
        gy = gy.reshape(1, E * gy.shape[1], *gy.shape[2:])
    # split-K: the reduction runs over batch x output pixels while the result is only Cin*kh*kw x Cout, so a layer
    # with a large feature map would occupy a handful of workgroups.  Batch chunks become extra "draws", summed after.
    E1, Bt = x.shape[0], x.shape[1]
    tiles = E1 * -(-Cin * kh * kw // 64) * -(-Cout // 64)
    split = 1
    while Bt % (2 * split) == 0 and Bt // (2 * split) >= 4 and tiles * 2 * split <= 1024:
        split *= 2
    if split > 1:
        x = x.reshape(E1 * split, Bt // split, *x.shape[2:])
        gy = gy.reshape(E1 * split, Bt // split, *gy.shape[2:])
    x_t = x.transpose(1, 2).contiguous()                                 # [E, Cin, B, H, W]
    gy_t = gy.transpose(1, 2).contiguous()                               # [E, Cout, B, Ho, Wo]
    g = conv2d_forward(x_t, gy_t, None, (dh, dw), (ph, pw), (sh, sw))    # [E, Cin, Cout, kh', kw'], kh' >= kh
    g = g[:, :, :, :kh, :kw]
    if split > 1:
        g = g.reshape(E1, split, *g.shape[1:]).sum(1)
    return g.transpose(1, 2).contiguous()


class _Conv2d(torch.autograd.Function):
    """y = conv2d(x, w, bias) batched over draws.  Backward runs on the same HIP GEMM (dgrad as a flipped-weight conv of the
    stride-upsampled gradient, wgrad as the axis-swapped conv)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation):
        y = conv2d_splitk(x, w, bias, stride, padding, dilation)
        ctx.save_for_backward(x, w)
        ctx.geom = (_pair(stride), _pair(padding), _pair(dilation), bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, has_bias = ctx.geom
        gy = gy.contiguous()
        E = gy.shape[0]
        gx = gw = gb = None
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(1, 3, 4))
            if w.shape[0] == 1 and E > 1:
                gb = gb.sum(0, keepdim=True)
        if ctx.needs_input_grad[1]:
            gw = conv2d_weight_grad(gy, x, tuple(w.shape), stride, padding, dilation)
        if ctx.needs_input_grad[0]:
            gx = conv2d_input_grad(gy, w, (E,) + tuple(x.shape[1:]), stride, padding, dilation)
            if x.shape[0] == 1 and E > 1:
                gx = gx.sum(0, keepdim=True)
        return gx, gw, gb, None, None, None


class _LrtConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_mu, w_var, b_mu, b_var, cfg):
        geom = (cfg["stride"], cfg["padding"], cfg["dilation"])
        if _small_launch(x, w_mu.unsqueeze(0), *geom):
            # training-sized launch: the fused kernel would run a few dozen workgroups with long k loops; compute the two
            # moments with split-K launches and sample separately (same noise stream, same result up to summation order)
            am = conv2d_splitk(x, w_mu.unsqueeze(0), None if b_mu is None else b_mu.unsqueeze(0), *geom)
            av = conv2d_splitk(x * x, w_var.unsqueeze(0), None if b_var is None else b_var.unsqueeze(0), *geom) + 1e-16
            if not cfg["sample"]:
                y = am
            elif cfg.get("eps") is not None:
                y = am + av.sqrt() * cfg["eps"].reshape(am.shape)
            else:
                y = lrt_sample_nchw(am, av, cfg["seed"], cfg["call0"], cfg["stream_id"])
        else:
            y, am, av = lrt_conv2d_forward(x, w_mu, w_var, b_mu, b_var, cfg["seed"], cfg["call0"], cfg["stream_id"], *geom,
                                           sample=cfg["sample"], eps=cfg.get("eps"), want_moments=True)
        ctx.cfg = cfg
        ctx.save_for_backward(x, w_mu, w_var, y, am, av)
        ctx.has_bias = b_mu is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w_mu, w_var, y, am, av = ctx.saved_tensors
        cfg = ctx.cfg
        geom = (cfg["stride"], cfg["padding"], cfg["dilation"])
        gy = gy.contiguous()
        # out = am + sqrt(av) * eps  ->  d/d am = g ;  d/d av = g * eps / (2 sqrt(av)) = g * (y - am) / (2 av)
        gv = gy * (y - am) / (2.0 * av) if cfg["sample"] else torch.zeros_like(gy)
        x2 = x * x
        wm5, wv5 = w_mu.unsqueeze(0), w_var.unsqueeze(0)
        gwm = conv2d_weight_grad(gy, x, tuple(wm5.shape), *geom)[0]
        gwv = conv2d_weight_grad(gv, x2, tuple(wv5.shape), *geom)[0]
        gx = None
        if ctx.needs_input_grad[0]:
            g1 = conv2d_input_grad(gy, wm5, tuple(x.shape), *geom)
            g2 = conv2d_input_grad(gv, wv5, tuple(x.shape), *geom)
            gx = g1 + g2 * 2.0 * x
        gbm = gy.sum(dim=(0, 1, 3, 4)) if ctx.has_bias else None
        gbv = gv.sum(dim=(0, 1, 3, 4)) if ctx.has_bias else None
        return gx, gwm, gwv, gbm, gbv, None


def _no_grad_needed(tensors):
    return not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))


def sample_weights(mus, rhos, prior_mu, prior_sigma, stream_ids, seed, call0, draws=1, eps=None, textbook_kl=False):
    if _no_grad_needed(list(mus) + list(rhos)):
        # inference: the launch itself -- autograd.Function.apply on 24 tensors costs ~80 us of host time per call
        ws, _, kl = reparam_kl_forward([m.detach() for m in mus], [r.detach() for r in rhos], prior_mu, prior_sigma, list(stream_ids), seed,
                                       call0, draws, sample=True, eps=eps, textbook_kl=textbook_kl)
        return kl, ws
    cfg = dict(prior_mu=prior_mu, prior_sigma=prior_sigma, stream_ids=list(stream_ids), seed=seed, call0=call0,
               draws=draws, eps=eps, textbook_kl=textbook_kl)
    flat = []
    for m, r in zip(mus, rhos):
        flat += [m, r]
    out = _SampleWeights.apply(cfg, *flat)
    return out[0], list(out[1:])


def kl_only(mus, rhos, prior_mu, prior_sigma, want_sigma=False, sigma_squared=False, textbook_kl=False, mu_through=False):
    """-> (kl, [sigma ...]) -- with mu_through (and want_sigma) -> (kl, [sigma ...], [mu ...]): the means as outputs of the same
    autograd node, so that gradients w.r.t. them are added to d KL / d mu inside its one backward launch."""
    if _no_grad_needed(list(mus) + list(rhos)):
        _, sig, kl = reparam_kl_forward([m.detach() for m in mus], [r.detach() for r in rhos], prior_mu, prior_sigma, [0] * len(mus), 0, 0, 1,
                                        sample=False, want_sigma=want_sigma, sigma_squared=sigma_squared, textbook_kl=textbook_kl)
        sig = list(sig) if sig is not None else []
        return (kl, sig, list(mus)) if mu_through else (kl, sig)
    cfg = dict(prior_mu=prior_mu, prior_sigma=prior_sigma, stream_ids=[0] * len(mus), want_sigma=want_sigma,
               sigma_squared=sigma_squared, textbook_kl=textbook_kl, mu_through=mu_through)
    flat = []
    for m, r in zip(mus, rhos):
        flat += [m, r]
    out = _KLOnly.apply(cfg, *flat)
    if mu_through:
        n_sig = len(out) - 1 - len(mus)
        return out[0], list(out[1:1 + n_sig]), list(out[1 + n_sig:])
    return out[0], list(out[1:])


def conv2d(x, w, bias, stride=1, padding=0, dilation=1):
    """Differentiable conv2d batched over draws.  Without autograd (inference through the drop-in layers) this is the
    plain kernel launch, whose results the batched ensemble path reproduces bit for bit; with autograd the split-K
    variant may be used (training launches are small)."""
    if not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad))):
        return conv2d_forward(x, w, bias, stride, padding, dilation)
    return _Conv2d.apply(x, w, bias, stride, padding, dilation)


def lrt_conv2d(x, w_mu, w_var, b_mu, b_var, seed, call0, stream_id, stride=1, padding=0, dilation=1, sample=True, eps=None):
    cfg = dict(seed=seed, call0=call0, stream_id=stream_id, stride=stride, padding=padding, dilation=dilation,
               sample=sample, eps=eps)
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w_mu, w_var, b_mu, b_var))
    if not needs_grad:          # inference through the drop-in layers: the fused kernel, bitwise what the ensemble path runs
        return lrt_conv2d_forward(x, w_mu, w_var, b_mu, b_var, seed, call0, stream_id, stride, padding, dilation,
                                  sample=sample, eps=eps)[0]
    return _LrtConv2d.apply(x, w_mu, w_var, b_mu, b_var, cfg)


# ------------------------------------------------------------------------------------------------
# batch-innermost backward helpers (training on the fast path, bbb_hip/fast_train.py)
# ------------------------------------------------------------------------------------------------
def padded_plane_pitch(K):
    """Row pitch for a [planes, K] matrix that a GEMM reads row-wise: K itself unless rows of K floats are a multiple of 4 KiB
    apart (then every row of a 64-row tile would start in the same memory channel: measured 10x slower), else K + 32."""
    return K + 32 if K % 1024 == 0 else K


def pool_act_backward_chwn(g_out, y, k, s, act, pad_planes=False):
    """Backward of [fused activation -> MaxPool2d(k, s)] (k = 0: activation only) on [..., H, W, B] planes: gradient w.r.t.
    the layer's pre-activation from the gradient w.r.t. the (pooled) output and the ACTIVATED output y (bbb_pool_act_bwd_chwn).
    pad_planes: return a view of y's shape into a [planes, padded_plane_pitch(H*W*B)] buffer (see there)."""
    require_device(g_out, y)
    g_out, y = g_out.contiguous(), y.contiguous()
    *lead, H, W, B = y.shape
    planes = 1
    for v in lead:
        planes *= v
    K = H * W * B
    pitch = padded_plane_pitch(K) if pad_planes else K
    if pitch != K:
        buf = torch.empty((planes, pitch), dtype=torch.float32, device=y.device)
        g_pre = buf[:, :K].view(*lead, H, W, B)
    else:
        buf = g_pre = torch.empty_like(y)
    with on_device(y.device):
        check(_lib.lib().bbb_pool_act_bwd_chwn(g_out.data_ptr(), y.data_ptr(), buf.data_ptr(), planes, H, W, B, int(k), int(s),
                                               ACT_CODE[act], pitch if pitch != K else 0, cur_stream(y.device)), "bbb_pool_act_bwd_chwn")
    return g_pre


def plane_sums(g, over_draws=False):
    """Bias gradient from the gradient w.r.t. a layer's pre-activation g [E, C, H, W, B] (possibly a padded-pitch view as
    pool_act_backward_chwn returns it): -> [E, C] (sum over pixels and images), or [C] with over_draws (bbb_plane_sum)."""
    require_device(g)
    E, C, H, W, B = g.shape
    cols = H * W * B
    if g.stride(4) != 1 or g.stride(3) != B or g.stride(2) != W * B or g.stride(0) != C * g.stride(1):
        g = g.contiguous()
    pitch = g.stride(1)
    out = torch.empty((C,) if over_draws else (E, C), dtype=torch.float32, device=g.device)
    with on_device(g.device):
        if over_draws:
            check(_lib.lib().bbb_plane_sum(g.data_ptr(), out.data_ptr(), E, C, cols, pitch, C * pitch, cur_stream(g.device)), "bbb_plane_sum")
        else:
            check(_lib.lib().bbb_plane_sum(g.data_ptr(), out.data_ptr(), 1, E * C, cols, pitch, 0, cur_stream(g.device)), "bbb_plane_sum")
    return out


def sum_over_draws(x, keepdim=False):
    """x [E, ...] -> the sum over the leading (draw) axis, added in draw order (bbb_sum_leading); trailing size % 4 == 0."""
    require_device(x)
    x = x.contiguous()
    E = x.shape[0]
    if E == 1:
        return x if keepdim else x[0]
    n = x.numel() // E
    out = torch.empty(((1,) if keepdim else ()) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    with on_device(x.device):
        check(_lib.lib().bbb_sum_leading(x.data_ptr(), out.data_ptr(), E, n, cur_stream(x.device)), "bbb_sum_leading")
    return out


def square(x, out=None):
    """x * x (bbb_lrt_glue mode 0): the operand of the LRT variance contraction's weight gradient.  out: a contiguous tensor of
    x's element count to write into."""
    require_device(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    elif not out.is_contiguous() or out.numel() != x.numel() or out.dtype != torch.float32:
        raise _lib.BBBHipError("square: out must be a contiguous fp32 tensor of x's size")
    with on_device(x.device):
        check(_lib.lib().bbb_lrt_glue(0, x.data_ptr(), 0, out.data_ptr(), x.numel(), 0, 0, cur_stream(x.device)), "bbb_lrt_glue")
    return out


def lrt_input_grad_combine(g1, x, g2):
    """g1 + 2 * x * g2 with x [1|E, ...] broadcast over the draws of g1 / g2 [E, ...] (bbb_lrt_glue mode 1)."""
    require_device(g1, x, g2)
    g1, x, g2 = g1.contiguous(), x.contiguous(), g2.contiguous()
    out = torch.empty_like(g1)
    with on_device(g1.device):
        check(_lib.lib().bbb_lrt_glue(g1.data_ptr(), x.data_ptr(), g2.data_ptr(), out.data_ptr(), g1.numel(), x.numel(), 1,
                                      cur_stream(g1.device)), "bbb_lrt_glue")
    return out


def lrt_pool_act_backward_chwn(g_out, y, act_mu, act_var, k, s, act, pad_planes=False, stacked=False, combine=None):
    """pool_act_backward_chwn for a local-reparameterisation layer: -> (g_mu, g_var), the gradients w.r.t. act_mu and act_var
    (bbb_lrt_pool_act_bwd_chwn).  act_mu / act_var: y's shape, or one draw's worth ([1, C, H, W, B]) when every draw was sampled
    from the same pair of moments.  stacked (without pad_planes): the two land in ONE buffer [2, *y.shape], returned as such --
    the weight-side and input-side contractions of the pair then run as the draws of one launch.
    combine = (x_out, g_out2): the incoming gradient is g_out + 2 * x_out * g_out2 (lrt_input_grad_combine of the layer above,
    formed on the fly: g_out / g_out2 = its two input gradients, x_out = this layer's output, [E|1, ...] of g_out's trailing shape)."""
    require_device(g_out, y, act_mu, act_var)
    g_out, y, act_mu, act_var = g_out.contiguous(), y.contiguous(), act_mu.contiguous(), act_var.contiguous()
    g2p, xcp, x_planes = 0, 0, 0
    if combine is not None:
        xc, g2 = combine[0].contiguous(), combine[1].contiguous()
        require_device(xc, g2)
        if g2.numel() != g_out.numel() or g_out.numel() % xc.numel() != 0:
            raise _lib.BBBHipError("lrt_pool_act_backward_chwn: combine = (x_out, g_out2) of g_out's shape (x_out may hold one draw)")
        g2p, xcp = g2.data_ptr(), xc.data_ptr()
    *lead, H, W, B = y.shape
    planes = 1
    for v in lead:
        planes *= v
    mom_planes = act_mu.numel() // (H * W * B)
    if act_mu.shape != act_var.shape or act_mu.numel() != mom_planes * H * W * B or planes % max(mom_planes, 1) != 0:
        raise _lib.BBBHipError("lrt_pool_act_backward_chwn: act_mu / act_var must hold y's planes, or one draw's worth of them")
    K = H * W * B
    pitch = padded_plane_pitch(K) if pad_planes else K
    if pitch != K:
        bufs = [torch.empty((planes, pitch), dtype=torch.float32, device=y.device) for _ in range(2)]
        outs = [b[:, :K].view(*lead, H, W, B) for b in bufs]
    elif stacked:
        both = torch.empty((2,) + tuple(y.shape), dtype=torch.float32, device=y.device)
        bufs = outs = [both[0], both[1]]
    else:
        bufs = outs = [torch.empty_like(y), torch.empty_like(y)]
    with on_device(y.device):
        check(_lib.lib().bbb_lrt_pool_act_bwd_chwn(g_out.data_ptr(), y.data_ptr(), act_mu.data_ptr(), act_var.data_ptr(),
                                                   bufs[0].data_ptr(), bufs[1].data_ptr(), planes, mom_planes, H, W, B, int(k), int(s),
                                                   ACT_CODE[act], pitch if pitch != K else 0, g2p, xcp,
                                                   (planes * xc.numel() // g_out.numel()) if combine is not None else 0,
                                                   cur_stream(y.device)),
              "bbb_lrt_pool_act_bwd_chwn")
    if stacked and pitch == K:
        return both
    return outs[0], outs[1]


def flip_transpose_w(w):
    """[E, Cout, Cin, kh, kw] -> [E, Cin, Cout, kh, kw] with the taps spatially flipped: the weight operand of the input gradient."""
    w = w.contiguous()
    E, Cout, Cin, kh, kw = w.shape
    w_t = torch.empty((E, Cin, Cout, kh, kw), dtype=torch.float32, device=w.device)
    with on_device(w.device):
        check(_lib.lib().bbb_flip_transpose_w(w.data_ptr(), w_t.data_ptr(), E, Cout, Cin, kh * kw, cur_stream(w.device)),
              "bbb_flip_transpose_w")
    return w_t


def flip_transpose_w_pair(w0, w1):
    """flip_transpose_w of two weight sets of one shape as ONE operand: [E, Cout, Cin, kh, kw] x 2 -> [2E, Cin, Cout, kh, kw], w0's
    draws first (bbb_flip_transpose_w_pair: an LRT layer's mean and variance weights for both input gradients in one launch)."""
    require_device(w0, w1)
    w0, w1 = w0.contiguous(), w1.contiguous()
    if w0.shape != w1.shape or w0.dim() != 5:
        raise _lib.BBBHipError("flip_transpose_w_pair: two [E, Cout, Cin, kh, kw] tensors of one shape")
    E, Cout, Cin, kh, kw = w0.shape
    w_t = torch.empty((2 * E, Cin, Cout, kh, kw), dtype=torch.float32, device=w0.device)
    with on_device(w0.device):
        check(_lib.lib().bbb_flip_transpose_w_pair(w0.data_ptr(), w1.data_ptr(), w_t.data_ptr(), E, Cout, Cin, kh * kw,
                                                   cur_stream(w0.device)), "bbb_flip_transpose_w_pair")
    return w_t


def flip_transpose_w_multi(sets):
    """flip_transpose_w / flip_transpose_w_pair for several layers in ONE launch (bbb_flip_transpose_w_multi): `sets` = a list of
    w [E, Cout, Cin, kh, kw] or (w0, w1) pairs of one shape -> the list of [E | 2E, Cin, Cout, kh, kw] results."""
    if not sets:
        return []
    if len(sets) > 16:
        return flip_transpose_w_multi(sets[:16]) + flip_transpose_w_multi(sets[16:])
    segs = (_lib.FlipSeg * len(sets))()
    outs, hold = [], []
    for i, s_ in enumerate(sets):
        w0, w1 = s_ if isinstance(s_, (tuple, list)) else (s_, None)
        require_device(w0) if w1 is None else require_device(w0, w1)
        w0 = w0.contiguous()
        E, Cout, Cin, kh, kw = w0.shape
        if w1 is not None:
            w1 = w1.contiguous()
            if w1.shape != w0.shape:
                raise _lib.BBBHipError("flip_transpose_w_multi: the two sources of a pair must have one shape")
        n = E if w1 is None else 2 * E
        o = torch.empty((n, Cin, Cout, kh, kw), dtype=torch.float32, device=w0.device)
        segs[i].w0, segs[i].w1, segs[i].out = w0.data_ptr(), (0 if w1 is None else w1.data_ptr()), o.data_ptr()
        segs[i].draws, segs[i].cout, segs[i].cin, segs[i].khkw = n, Cout, Cin, kh * kw
        outs.append(o)
        hold += [w0, w1]
    dev = outs[0].device
    with on_device(dev):
        check(_lib.lib().bbb_flip_transpose_w_multi(segs, len(sets), cur_stream(dev)), "bbb_flip_transpose_w_multi")
    return outs


def conv2d_chwn_input_grad(g_pre, w, x_hw, padding, dilation, w_flipped=None):
    """d loss / d x of a STRIDE-1 y = conv(x, w) in the batch-innermost layout, on the forward kernel itself: the convolution
    of g_pre [E, Cout, Ho, Wo, B] with the spatially flipped, channel-transposed weights, padding d*(k-1) - p.
    w [E, Cout, Cin, kh, kw] -> [E, Cin, H, W, B].  w_flipped: flip_transpose_w(w) computed ahead (off the gradient chain)."""
    (ph, pw), (dh, dw) = _pair(padding), _pair(dilation)
    kh, kw = w.shape[3], w.shape[4]
    qh, qw = dh * (kh - 1) - ph, dw * (kw - 1) - pw
    if qh < 0 or qw < 0:
        raise _lib.BBBHipError("conv2d_chwn_input_grad: padding larger than the kernel reach")
    w_t = w_flipped if w_flipped is not None else flip_transpose_w(w)
    gx = conv2d_chwn_forward(g_pre, w_t, None, 1, (qh, qw), (dh, dw))
    if gx.shape[2] != x_hw[0] or gx.shape[3] != x_hw[1]:
        raise _lib.BBBHipError("conv2d_chwn_input_grad: geometry mismatch (stride-1 layers only)")
    return gx


def _transpose_batched(src, out, rows, cols, nb1, nb2, ib1, ib2, ir, ob1, ob2, oc):
    nb = nb1 * nb2
    if nb > 65535 or (rows + 31) // 32 > 65535:
        return False
    with on_device(src.device):
        check(_lib.lib().bbb_transpose_batched(src.data_ptr(), out.data_ptr(), rows, cols, nb1, nb2, ib1, ib2, ir, ob1, ob2, oc,
                                               cur_stream(src.device)), "bbb_transpose_batched")
    return True


def _transpose_sum_batched(src, out, rows, cols, nb, ib, ob, ir, oc, nsum=1, in_sum=0, square_off=0):
    """bbb_transpose_sum_batched: three batch dimensions `nb` with strides `ib` / `ob`, `nsum` slices `in_sum` apart summed in
    ascending order; square_off != 0: the squares land square_off elements behind the outputs.  False (nothing launched) when the
    batch does not fit one grid."""
    if nb[0] * nb[1] * nb[2] > 65535 or (rows + 31) // 32 > 65535:
        return False
    with on_device(src.device):
        check(_lib.lib().bbb_transpose_sum_batched(src.data_ptr(), out.data_ptr(), rows, cols, (ctypes.c_int32 * 3)(*nb),
                                                   (ctypes.c_int64 * 3)(*ib), (ctypes.c_int64 * 3)(*ob), ir, oc, nsum, in_sum,
                                                   int(square_off), cur_stream(src.device)), "bbb_transpose_sum_batched")
    return True


def chwn_to_bhwc(x, out=None):
    """[E, C, H, W, B] -> [E, B, H, W, C] (the batch becomes the contraction channels, C the innermost axis).  out: a contiguous
    [E, B, H, W, C] tensor to write into."""
    E, C, H, W, B = x.shape
    x = x.contiguous()
    if out is None:
        out = torch.empty((E, B, H, W, C), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (E, B, H, W, C) or not out.is_contiguous():
        raise _lib.BBBHipError("chwn_to_bhwc: out must be a contiguous [E, B, H, W, C] tensor")
    HW = H * W
    if not _transpose_batched(x, out, C, B, E, HW, C * HW * B, B, HW * B, B * HW * C, C, HW * C):
        out.copy_(x.permute(0, 4, 2, 3, 1))
    return out


def chwn_grad_as_weights(g, chunks=1):
    """[E, Cout, Ho, Wo, B] -> [E, Cout, B, Ho, Wo] (the output gradient in the layout of a weight operand); with `chunks` = S
    the batch splits into S slices that become extra draws: [E*S, Cout, B/S, Ho, Wo], slice s of draw e at index e*S + s."""
    E, Co, Ho, Wo, B = g.shape
    g = g.contiguous()
    P_ = Ho * Wo
    S = int(chunks)
    if S > 1:
        Bs = B // S
        out = torch.empty((E * S, Co, Bs, Ho, Wo), dtype=g.dtype, device=g.device)
        if _transpose_sum_batched(g, out, P_, Bs, (E, S, Co), (Co * P_ * B, Bs, P_ * B), (S * Co * Bs * P_, Co * Bs * P_, Bs * P_),
                                  B, P_):
            return out
        return (chwn_grad_as_weights(g).reshape(E, Co, S, Bs, Ho, Wo).permute(0, 2, 1, 3, 4, 5)
                .reshape(E * S, Co, Bs, Ho, Wo))
    out = torch.empty((E, Co, B, Ho, Wo), dtype=g.dtype, device=g.device)
    if not _transpose_batched(g, out, P_, B, E * Co, 1, P_ * B, 0, B, B * P_, 0, P_):
        out = g.permute(0, 1, 4, 2, 3).contiguous()
    return out


# conv2d_chwn_weight_grad may read a large launch's output gradient in place as a tap-major weight operand (no transposed copy, 250 MB
# per 512 x 10 step).  Measured slower, so off: 2.54 against 2.47 ms per step (profiles/experiments/ab_wgrad_inplace.py) -- with the
# images innermost in k, consecutive rows of the role-swapped input are a whole image apart instead of adjacent.
wgrad_in_place = [False]


def conv2d_chwn_weight_grad(g_pre, x, w_shape, stride, padding, dilation, x_squares=False):
    """d loss / d w in the batch-innermost layout, again on the forward kernel with the roles swapped: the batch becomes the
    contraction channels, the layer's input channels the innermost ("image") axis, the output pixels the kernel taps:
        gw[e][n][tap][ci] = sum_{b, opix} g_pre'[e][n][b][opix] * x'[e][b][ipix(tap, opix)][ci]
    with x' = x as [E|1, B, H, W, Cin], g_pre' = g_pre as [E, Cout, B, Ho, Wo], stride <-> dilation swapped.  Padding taps
    are skipped as in the forward.  The innermost axis moves as 16-byte vectors: an input with Cin % 4 != 0 is padded with zero
    planes first.  x may be shared by all draws ([1, ...]).  A launch that would occupy fewer than 512 workgroups (one draw of a small model) splits the batch into
    S chunks that run as extra draws and are summed in a fixed order.  g_pre [E, Cout, Ho, Wo, B], x [E|1, Cin, H, W, B] ->
    [E, Cout, Cin, kh, kw].
    x_squares (the two weight gradients of a local-reparameterisation layer as one launch): g_pre holds 2E' draws -- the gradients
    w.r.t. the E' mean activations, then the E' variance activations -- and x [E', ...] pairs with the first half, its squares with
    the second; -> [2E', Cout, Cin, kh, kw]."""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ew, Cout, Cin, kh, kw = w_shape
    if Cin % 4 != 0:
        # the innermost axis of the role-swapped launch moves as 16-byte vectors: pad the input channels with zero planes
        # (BayesianLeNet's 6-channel second conv) and drop their -- all-zero -- gradient columns
        Cp = (Cin + 3) & ~3
        xp = x.new_zeros((x.shape[0], Cp) + tuple(x.shape[2:]))
        xp[:, :Cin] = x
        return conv2d_chwn_weight_grad(g_pre, xp, (Ew, Cout, Cp, kh, kw), stride, padding, dilation,
                                       x_squares=x_squares)[:, :, :Cin].contiguous()
    E, B = g_pre.shape[0], g_pre.shape[4]
    if x_squares:
        Ex = x.shape[0]
        if E != 2 * Ex:
            raise _lib.BBBHipError("conv2d_chwn_weight_grad(x_squares): g_pre must hold two draws per draw of x")
        xr = torch.empty((2 * Ex, B, x.shape[2], x.shape[3], Cin), dtype=torch.float32, device=x.device)
        H_, W_ = x.shape[2], x.shape[3]
        HW_ = H_ * W_
        xc = x.contiguous()
        # the transposed x and, behind it, its squares (x^2 transposed = the transposed x, squared) in ONE pass
        if not _transpose_sum_batched(xc, xr, Cin, B, (Ex, HW_, 1), (Cin * HW_ * B, B, 0), (B * HW_ * Cin, Cin, 0), HW_ * B, HW_ * Cin,
                                      square_off=Ex * B * HW_ * Cin):
            chwn_to_bhwc(xc, out=xr[:Ex])
            square(xr[:Ex], out=xr[Ex:])
    else:
        xr = chwn_to_bhwc(x)                                            # [E|1, B, H, W, Cin]
    wgs = E * kh * kw * -(-Cout // 64) * -(-Cin // 64)
    S = 1
    while wgs * S < 512 and B % (2 * S) == 0 and B // (2 * S) >= 8:
        S *= 2
    if S == 1 and wgrad_in_place[0] and current_config().gemm_mode != "bf16x3":
        # the output gradient [E, Cout, Ho, Wo, B] IS a tap-major weight operand (taps = output pixels, channels = images): read in
        # place, the contraction running (pixel, image) instead of (image, pixel) -- no transposed copy (250 MB per 512 x 10 step)
        y = conv2d_chwn_forward(xr, g_pre.contiguous(), None, (dh, dw), (ph, pw), (sh, sw), w_tap_major=True)
    else:
        gr = chwn_grad_as_weights(g_pre, S)                             # [E*S, Cout, B/S, Ho, Wo]
        if S > 1:
            if xr.shape[0] == 1 and E > 1:
                xr = xr.expand(E, *xr.shape[1:])
            xr = xr.reshape(xr.shape[0] * S, B // S, *xr.shape[2:])
        y = conv2d_chwn_forward(xr, gr, None, (dh, dw), (ph, pw), (sh, sw))  # [E*S, Cout, kh', kw', Cin], kh' >= kh
    if y.shape[2] == kh and y.shape[3] == kw:                           # sum_s [E, S, Cout, kh*kw, Cin] -> [E, Cout, Cin, kh*kw]
        gw = torch.empty((E, Cout, Cin, kh, kw), dtype=torch.float32, device=y.device)
        T = kh * kw
        if _transpose_sum_batched(y, gw, T, Cin, (E, Cout, 1), (S * Cout * T * Cin, T * Cin, 0), (Cout * Cin * T, Cin * T, 0), Cin, T,
                                  S, Cout * T * Cin):
            return gw
    if S > 1:
        y = y.reshape(E, S, *y.shape[1:]).sum(1)
    return y[:, :, :kh, :kw, :].permute(0, 1, 4, 2, 3).contiguous()


def im2col_pbj(x_nchw, w_shape, stride, padding, dilation):
    """The im2col of a shared first-layer input for conv2d_chwn_weight_grad_shared_input (bbb_im2col_pbj): x [B, Cin, H, W] ->
    [Ho * Wo, B, Jp] with j = (ci, r, q), Jp = Cin * kh * kw rounded up to 4 (zero columns).  It depends on the batch alone: a
    training step may build it off the gradient chain; its elementwise square is the im2col of x^2 (the LRT variance side)."""
    _, Cout, Cin, kh, kw = w_shape
    x_nchw = x_nchw.contiguous()
    dd, ho, wo = _desc(x_nchw.unsqueeze(0), _Shape((1, Cout, Cin, kh, kw)), stride, padding, dilation, 1, False, False, None)
    Jp = (Cin * kh * kw + 3) // 4 * 4
    xk = torch.empty((ho * wo, x_nchw.shape[0], Jp), dtype=torch.float32, device=x_nchw.device)
    with on_device(x_nchw.device):
        check(_lib.lib().bbb_im2col_pbj(x_nchw.data_ptr(), xk.data_ptr(), ctypes.byref(dd), cur_stream(x_nchw.device)), "bbb_im2col_pbj")
    return xk


def conv2d_chwn_weight_grad_shared_input(g_pre, x_nchw, w_shape, stride, padding, dilation, xk=None):
    """Weight gradient of a layer whose INPUT is the same for every draw (a model's first layer; 3-channel images do not fit
    the role-swapped launch, whose innermost axis would be Cin).  The draws stack into the row dimension instead:
        GW[(e, n)][j] = sum_k G[(e, n)][k] * Xcol[k][j],   k = (output pixel, image), j = (ci, r, q)
    -- g_pre's own memory IS the [E*Cout, K] matrix G, Xcol is the im2col of the shared input (built once: F.unfold + one
    permute), and the product runs on the forward kernel as a 1x1 "convolution" with K as the contraction channels and j as
    the innermost axis, K split into S slices that run as the launch's draws (w_row_pitch lets a slice of G's rows be read in
    place) and are summed in a fixed order.  g_pre [E, Cout, Ho, Wo, B], x_nchw [B, Cin, H, W] -> [E, Cout, Cin, kh, kw].
    xk: im2col_pbj(x_nchw, ...) built ahead (x_nchw is then only checked for its batch size and may be None)."""
    E, Cout, Ho, Wo, B = g_pre.shape
    _, _, Cin, kh, kw = w_shape
    J, P_ = Cin * kh * kw, Ho * Wo
    Jp = (J + 3) // 4 * 4
    K = P_ * B
    if xk is None:
        xk = im2col_pbj(x_nchw, w_shape, stride, padding, dilation)
    if tuple(xk.shape) != (P_, B, Jp) or not xk.is_contiguous():
        raise _lib.BBBHipError("conv2d_chwn_weight_grad_shared_input: geometry mismatch")
    M = E * Cout
    wgs = -(-M // 64) * -(-Jp // 128)            # slices sized for the launcher's 128-wide tiles (>= 768 of them; 64-wide: 2.50 -> 2.48 ms per 512 x 10 step)
    S = 1
    while wgs * S < 768 and K % (2 * S) == 0 and (K // (2 * S)) % 4 == 0 and K // (2 * S) >= 256:
        S *= 2
    Ks = K // S
    d = ConvDesc()
    d.batch, d.cin, d.h, d.w, d.cout, d.kh, d.kw = Jp, Ks, 1, 1, M, 1, 1
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1
    d.pad_h = d.pad_w = 0
    d.draws = S
    d.x_draw_stride = Ks * Jp
    d.w_draw_stride = Ks
    d.b_draw_stride = 0
    d.act = 0
    # G's rows must not be a multiple of 4 KiB apart (128 KiB for 64 pixels x 512 images: the 64 rows of a weight tile would
    # all sit in the same memory channel).  pool_act_backward_chwn(pad_planes=True) already wrote g_pre at the padded pitch;
    # anything else is copied once.
    pitch = padded_plane_pitch(K)
    if g_pre.is_contiguous() and pitch == K:
        gp = g_pre
    elif g_pre.stride(1) == pitch and g_pre.stride(0) == Cout * pitch and g_pre[0, 0].is_contiguous():
        gp = g_pre                                                     # a view into the padded [M, pitch] buffer
    else:
        gp = torch.empty((M, pitch), dtype=torch.float32, device=g_pre.device)
        gp[:, :K] = g_pre.reshape(M, K)
    d.w_row_pitch = pitch
    y = torch.empty((S, M, Jp), dtype=torch.float32, device=g_pre.device)
    with on_device(g_pre.device):
        check(_lib.lib().bbb_conv2d_chwn_fwd(ctypes.byref(d), xk.data_ptr(), gp.data_ptr(), 0, y.data_ptr(),
                                             cur_stream(g_pre.device)), "bbb_conv2d_chwn_fwd")
    gw = y.sum(0) if S > 1 else y[0]
    return gw[:, :J].reshape(E, Cout, Cin, kh, kw)


class _OpsModule(types.ModuleType):
    """`ops.gemm_mode`, `ops.split_k`, ... read and write the fields of the process-default LaunchConfig (compatibility with the
    round-1..4 switches; prefer `use_config` / `launch_config=`)."""

    def __getattr__(self, name):                     # only reached when the module dict has no such name
        if name in LaunchConfig.FIELDS:
            return getattr(_default_config, name)
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        if name in LaunchConfig.FIELDS:
            setattr(_default_config, name, value)
        else:
            super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OpsModule
