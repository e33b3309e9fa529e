"""The num_ens Monte-Carlo loop of the reference, batched over draws and sharded over GPUs.

Reference semantics (main_bayesian.py:43-53 train, :73-80 validate):
    for j in range(num_ens): net_out, _kl = net(x); kl += _kl; outputs[:, :, j] = log_softmax(net_out, 1)
    log_outputs = utils.logmeanexp(outputs, dim=2)
Here the E draws run as ONE pass: one fused reparam+KL launch materialises the E weight sets of every
layer (draw j uses noise call index call0 + j, so the result equals the Python loop over `net(x)`), each
conv / linear is one launch batched over draws, and one tail kernel does log_softmax + log-sum-exp over
draws.  Multi-GPU (strong scaling): the work grid (draw j) x (batch slice s) is enumerated
draw-major, u = j*S + s, and dealt out in contiguous equal ranges; parameters are replicated and the noise is keyed by
(seed, call = draw, stream, element), so a rank materialises exactly the weight sets its units touch -- no weight
traffic.  Each rank reduces its units to a [B, C] log-sum-exp block (-inf where it holds no unit of a slice), and ONE
all_gather of [B*C + 1] floats (the block + its share of the KL sum) over RCCL is followed by a log-sum-exp over ranks
in rank order -- every rank ends with the same bits.
"""
import math
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, rng, _lib

try:
    from layers.bbb import _BBBLayer, BBBConv2d as _BBBConv, BBBLinear as _BBBLin
    from layers.lrt import _LRTLayer, BBBConv2d as _LRTConv, BBBLinear as _LRTLin
    from layers.misc import FlattenLayer
except ImportError as e:  # pragma: no cover
    raise ImportError("bbb_hip.ensemble needs the sibling `layers` package on sys.path") from e


fast_autograd = True      # False: training forwards take the reference-layout autograd path (tests compare the two)
stats = {"path": None, "launch": None}   # which layout the last mc_logits / mc_forward took: "chwn" (batch-innermost fast path) or "nchw"


def draw_range(num_ens, rank, world):
    """Contiguous, balanced block of global draw indices for `rank`: [lo, hi)."""
    base, rem = divmod(num_ens, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_slices(num_ens, world, batch, multiple=4, min_slice=64):
    """Batch slices per draw (S) for `world` ranks: the smallest S in {1, 2, 4, 8, 16} whose busiest rank processes
    (within 5 %) the fewest images -- E=10 on 8 ranks: S=1 -> 2 draws x 512 = 1024 images on the busiest rank, S=4 -> 5 units
    x 128 = 640 = a perfect split.  Slices must keep `multiple` images alignment (4 fp32, 8 bf16) and at least `min_slice`
    images (smaller GEMM tiles waste the matrix cores)."""
    best = None
    for S in (1, 2, 4, 8, 16):
        if batch % S or (batch // S) % multiple or (S > 1 and batch // S < min_slice):
            continue
        cost = -(-num_ens * S // world) * (batch // S)
        if best is None or cost < 0.95 * best[0]:
            best = (cost, S)
    return 1 if best is None else best[1]


def unit_range(num_ens, slices, rank, world):
    """This rank's contiguous range [lo, hi) of the draw-major unit enumeration u = draw * slices + slice."""
    return draw_range(num_ens * slices, rank, world)


_epoch = [0]     # bumped whenever ANY nn.Module registers a sub-module or a parameter (torch's global registration hooks)


def _bump_epoch(*_args):
    _epoch[0] += 1
    return None


try:
    import torch.nn.modules.module as _tm
    _tm.register_module_module_registration_hook(_bump_epoch)
    _tm.register_module_parameter_registration_hook(_bump_epoch)
except AttributeError:  # pragma: no cover  (a torch without the global registration hooks: direct children only)
    pass


def _structure(net):
    """Static analysis of `net`, cached on the module: the answers below are asked on every forward of a drop-in loop
    (`for j in range(E): net(x)`), and walking the module tree each time cost more host time than launching the kernels.
    The cache is dropped when net's direct children change identity or when any module anywhere registers a sub-module or
    a parameter after the analysis (`net.classifier[2] = BBBLinear(...)`, `layer.W_mu = nn.Parameter(...)`: torch's global
    registration hooks bump an epoch).  Only direct surgery on `_modules` / `_parameters` dicts needs invalidate(net)."""
    sig = tuple(map(id, net._modules.values()))
    st = net.__dict__.get("_bbb_structure")
    if st is None or st["sig"] != sig or st["epoch"] != _epoch[0]:
        st = {"sig": sig, "epoch": _epoch[0], "layers": [m for m in net.modules() if isinstance(m, (_BBBLayer, _LRTLayer))],
              "params": list(net.parameters()), "rows": {}, "train": {}}
        net.__dict__["_bbb_structure"] = st
    return st


def invalidate(net):
    """Forget the cached structure of `net` (after replacing modules inside one of its nested containers)."""
    net.__dict__.pop("_bbb_structure", None)


def bayesian_layers(net):
    return _structure(net)["layers"]


def any_requires_grad(net):
    return any(p.requires_grad for p in _structure(net)["params"])


def flat_children(net):
    """The model as the flat list of modules ModuleWrapper.forward runs: nn.Sequential and plain ModuleWrapper
    containers (no forward of their own, like the reference's models) are expanded recursively.  Returns None if a
    Bayesian layer sits inside any other kind of module -- the batched paths cannot see through an arbitrary forward."""
    st = _structure(net)
    if "flat" in st:
        return st["flat"]
    from layers.misc import ModuleWrapper
    out = []

    def walk(mod):
        for child in mod.children():
            plain_wrapper = isinstance(child, ModuleWrapper) and type(child).forward is ModuleWrapper.forward \
                and not isinstance(child, (_BBBLayer, _LRTLayer))
            if isinstance(child, nn.Sequential) or plain_wrapper:
                walk(child)
            else:
                out.append(child)

    walk(net)
    direct = {id(m) for m in out if isinstance(m, (_BBBLayer, _LRTLayer))}
    if direct != {id(m) for m in st["layers"]}:
        out = None
    st["flat"] = out
    return out


def output_rows(net, x_shape):
    """Rows of the model's output for an input of shape x_shape (B, except where FlattenLayer cuts a larger feature map into
    several rows per image: the reference's view(-1, num_features), layers/misc.py:35)."""
    cache = _structure(net)["rows"]
    if x_shape in cache:
        return cache[x_shape]
    B, C, H, W = x_shape
    rows = B
    for m in flat_children(net) or []:
        if isinstance(m, (_BBBConv, _LRTConv)):
            (sh, sw), (ph, pw), (dh, dw) = ops._pair(m.stride), ops._pair(m.padding), ops._pair(m.dilation)
            kh, kw = m.kernel_size
            H = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
            W = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
            C = m.out_channels
        elif isinstance(m, nn.MaxPool2d):
            k, st = m.kernel_size, m.stride
            H, W = (H - k) // st + 1, (W - k) // st + 1
        elif isinstance(m, FlattenLayer):
            rows = rows * C * H * W // m.num_features
            C, H, W = m.num_features, 1, 1
    cache[tuple(x_shape)] = rows
    return rows


class Timers:
    """Optional HIP-event brackets around individual kernel launches (bench.py roofline)."""

    def __init__(self):
        self.records = []   # (tag, info, start_event, end_event)

    def bracket(self, tag, info, fn):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((tag, info(out) if callable(info) else info, s, e))
        return out

    def summary(self):
        """tag -> {ms, n, work, work_im2col}: `work` = FLOPs actually required (in-bounds taps only) or elements,
        `work_im2col` = the 2*M*N*K of the zero-padded im2col GEMM (SURVEY.md section 8d figure)."""
        agg = {}
        for tag, info, s, e in self.records:
            a = agg.setdefault(tag, {"ms": 0.0, "n": 0, "work": 0.0, "work_im2col": 0.0})
            a["ms"] += s.elapsed_time(e)
            a["n"] += 1
            if isinstance(info, tuple):
                a["work"] += info[0]
                a["work_im2col"] += info[1]
            elif info is not None:
                a["work"] += info
                a["work_im2col"] += info
        return agg


def _run(timers, tag, info, fn):
    return timers.bracket(tag, info, fn) if timers is not None else fn()


def _sample_all(layers, draws, seed, call0, timers=None, eps=None, tm=()):
    """One fused launch (per <=16 tensors) for every BBB layer: -> ({layer: (w, b)}, kl).  tm: layers whose conv weights come out
    TAP-MAJOR, [draws, Cout, kh * kw, Cin] (ops.sample_weights_tm: the operand layout of ops.conv2d_c8x3_forward; inference)."""
    mus, rhos, ids, owners, tmf = [], [], [], [], []
    for l in layers:
        m, r, i = l._param_lists()
        mus += m
        rhos += r
        ids += i
        owners += [l] * len(m)
        tmf += [l in tm and t.dim() == 4 and t.shape[2] * t.shape[3] > 1 for t in m]
    pri = {(l.prior_mu, l.prior_sigma) for l in layers}
    out, kl_total = {}, None
    ws_all = []
    if len(pri) == 1:
        pm, ps = next(iter(pri))
        for s in range(0, len(mus), _lib.MAX_SEGMENTS):
            sl = slice(s, s + _lib.MAX_SEGMENTS)
            n_el = sum(m.numel() for m in mus[sl])
            if any(tmf[sl]):
                kl, ws = _run(timers, "reparam_kl", n_el,
                              lambda: ops.sample_weights_tm(mus[sl], rhos[sl], pm, ps, ids[sl], seed, call0, draws, tmf[sl]))
            else:
                kl, ws = _run(timers, "reparam_kl", n_el,
                              lambda: ops.sample_weights(mus[sl], rhos[sl], pm, ps, ids[sl], seed, call0, draws,
                                                         eps=None if eps is None else eps[sl]))
            kl_total = kl if kl_total is None else kl_total + kl
            ws_all += ws
    else:  # layers with different priors: one launch per layer
        k = 0
        for l in layers:
            m, r, i = l._param_lists()
            if any(tmf[k:k + len(m)]):
                kl, ws = ops.sample_weights_tm(m, r, l.prior_mu, l.prior_sigma, i, seed, call0, draws, tmf[k:k + len(m)])
            else:
                kl, ws = ops.sample_weights(m, r, l.prior_mu, l.prior_sigma, i, seed, call0, draws,
                                            eps=None if eps is None else eps[k:k + len(m)])
            k += len(m)
            kl_total = kl if kl_total is None else kl_total + kl
            ws_all += ws
    k = 0
    for l in layers:
        w = ws_all[k]
        b = ws_all[k + 1] if l.use_bias else None
        k += 2 if l.use_bias else 1
        out[l] = (w, b)
    return out, kl_total


def _sample_all_bf16(layers, draws, seed, call0, timers=None):
    """_sample_all for the bf16 storage path: weight matrices come back bf16 in the GEMM's padded-row layout."""
    mus, rhos, ids = [], [], []
    for l in layers:
        m, r, i = l._param_lists()
        mus += m
        rhos += r
        ids += i
    pri = {(l.prior_mu, l.prior_sigma) for l in layers}
    if len(pri) != 1:
        raise _lib.BBBHipError("bf16 path: all layers must share one prior")
    pm, ps = next(iter(pri))
    kl_total, ws_all = None, []
    for s in range(0, len(mus), _lib.MAX_SEGMENTS):
        sl = slice(s, s + _lib.MAX_SEGMENTS)
        kl, ws = _run(timers, "reparam_kl", sum(m.numel() for m in mus[sl]),
                      lambda: ops.sample_weights_bf16(mus[sl], rhos[sl], pm, ps, ids[sl], seed, call0, draws))
        kl_total = kl if kl_total is None else kl_total + kl
        ws_all += ws
    out, k = {}, 0
    for l in layers:
        out[l] = (ws_all[k], ws_all[k + 1] if l.use_bias else None)
        k += 2 if l.use_bias else 1
    return out, kl_total


def _variances_all(layers, timers=None):
    """One fused launch for every LRT layer: sigma^2 tensors + KL."""
    mus, rhos = [], []
    for l in layers:
        m, r, _ = l._param_lists()
        mus += m
        rhos += r
    pri = {(l.prior_mu, l.prior_sigma) for l in layers}
    if len(pri) != 1 or len(mus) > _lib.MAX_SEGMENTS:
        out, kl_total = {}, None
        for l in layers:
            m, r, _ = l._param_lists()
            kl, s2 = ops.kl_only(m, r, l.prior_mu, l.prior_sigma, want_sigma=True, sigma_squared=True)
            out[l] = (s2[0], s2[1] if l.use_bias else None)
            kl_total = kl if kl_total is None else kl_total + kl
        return out, kl_total
    pm, ps = next(iter(pri))
    kl, s2 = _run(timers, "reparam_kl", sum(m.numel() for m in mus),
                  lambda: ops.kl_only(mus, rhos, pm, ps, want_sigma=True, sigma_squared=True))
    out, k = {}, 0
    for l in layers:
        out[l] = (s2[k], s2[k + 1] if l.use_bias else None)
        k += 2 if l.use_bias else 1
    return out, kl


def _act_name(mod):
    if isinstance(mod, nn.ReLU):
        return "relu"
    if isinstance(mod, nn.Softplus) and mod.beta == 1 and mod.threshold == 20:
        return "softplus"
    return None


def conv_flops(B, Cin, H, W, Cout, kh, kw, stride, padding, dilation, draws, contractions=1):
    """(useful, im2col) FLOPs of a conv layer: useful counts only kernel taps that land inside the image."""
    sh, sw = ops._pair(stride)
    ph, pw = ops._pair(padding)
    dh, dw = ops._pair(dilation)
    ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    vr = sum(1 for o in range(ho) for r in range(kh) if 0 <= o * sh - ph + r * dh < H)
    vq = sum(1 for o in range(wo) for q in range(kw) if 0 <= o * sw - pw + q * dw < W)
    base = 2.0 * contractions * draws * B * Cout * Cin
    return base * vr * vq, base * ho * wo * kh * kw


def _chwn_ok(net, x, any_batch=True):
    """The batch-innermost fast path handles: 4-d input, no autograd, and only module kinds it knows how to run in that layout
    (Bayesian layers, ReLU/Softplus, MaxPool2d without padding, FlattenLayer that flattens whole images).  Batch sizes that are
    not a multiple of 4 (image rows move as 16-byte vectors) run PADDED with zero images whose rows are dropped again
    (_mc_logits_chwn); any_batch=False: only batches that need no padding (work units, several steps per launch)."""
    if torch.is_grad_enabled() and any_requires_grad(net):
        return False
    if x.dim() != 4 or x.shape[0] == 0 or (not any_batch and x.shape[0] % 4 != 0):
        return False
    st = _structure(net)
    if "chwn_mods" not in st:
        st["chwn_mods"] = _chwn_mods_ok(flat_children(net))
    return st["chwn_mods"]


def _chwn_mods_ok(mods):
    if mods is None:
        return False
    for m in mods:
        if isinstance(m, (_BBBLayer, _LRTLayer, FlattenLayer, nn.ReLU)):
            continue
        if isinstance(m, nn.Softplus) and m.beta == 1 and m.threshold == 20:
            continue
        if isinstance(m, nn.MaxPool2d):
            k, s = m.kernel_size, m.stride
            if isinstance(k, int) and isinstance(s, int) and m.padding == 0 and m.dilation == 1 and not m.ceil_mode:
                continue
        return False
    return True


def _check_precision(precision, net, x, fast_path_allowed):
    """precision: "fp32" (the reference's arithmetic, default); "bf16x3" (fp32 tensors and fp32 accuracy, the BBB GEMM launches of
    the batch-innermost path on the 16-bit matrix pipe -- range-free split bf16, ops.gemm_mode -- everything else as fp32); or
    "bf16" (bf16 storage of sampled weights and activations, fp32 accumulate; inference on the batch-innermost path only --
    anything else fails loudly)."""
    if precision in ("fp32", "bf16x3"):
        return
    if precision != "bf16":
        raise _lib.BBBHipError(f"precision must be 'fp32', 'bf16x3' or 'bf16', got {precision!r}")
    if not fast_path_allowed or not _chwn_ok(net, x, any_batch=False):
        raise _lib.BBBHipError("bf16 runs on the batch-innermost inference path only (no autograd, no external eps, "
                               "4-d input with B % 8 == 0, BBB layers + ReLU/Softplus/MaxPool2d/FlattenLayer)")


def _mc_logits_chwn(net, x, draws, seed, call0, timers=None, streams=1, precision="fp32", units=None, b_offset=0, groups=1,
                    share=None):
    """Inference path of mc_logits in the batch-innermost layout ([E, C, H, W, B]): pixel-major GEMMs that
    skip padding taps, activation fused into the GEMM epilogue, pooling on contiguous image vectors.
    units = (S, lo, hi): instead of `draws` whole draws starting at call0, run the work units lo..hi-1 of the draw-major
    (draw, batch slice) grid with S slices per draw (call0 = the call index of draw 0); returns logits [hi-lo, C, B/S].
    b_offset: global index of x's first image (batch-parallel shards): LRT activation noise is keyed by the global image.
    groups = G > 1: x holds G batches back to back ([G * B, C, H, W]) and the launches run G consecutive Monte-Carlo steps of
    `draws` forwards each -- slab g * draws + j = draw j of step g, on batch g, under noise call call0 + g * draws + j, i.e.
    exactly what G separate steps would compute (GraphedMC steps > 1); returns logits [G * draws, C, B].
    share = (D, off): one rank's part of a group of steps (group_share): `draws` consecutive slabs of the draw-major (step, draw)
    enumeration with D draws per step, the first being draw `off` of its step; x holds the ceil((draws + off) / D) batches they
    touch, call0 = the call index of the first local slab; returns logits [draws, C, B]."""
    layers = bayesian_layers(net)
    bbb = [l for l in layers if isinstance(l, _BBBLayer)]
    lrt = [l for l in layers if isinstance(l, _LRTLayer)]
    kl, sampled, variances = None, {}, {}
    bf16 = precision == "bf16"
    E, B = draws, x.shape[0]
    ukw = {}
    G = int(groups)
    rows_real = None
    if B % 4 != 0 and not bf16 and G == 1 and share is None and (units is None or units[0] <= 1):
        # an odd batch stays on the batch-innermost kernels: zero images fill it up to the next multiple of 4 (every image is
        # its own GEMM column -- BBB weights are shared, LRT noise is keyed by the global image index -- so the real images'
        # results are those of an unpadded run), and their output rows (they come last, also behind a flatten that cuts images
        # into several rows) are dropped again.  One small copy instead of the ~2x slower reference-layout kernels.
        rows_real = output_rows(net, tuple(x.shape))
        x = torch.cat([x, x.new_zeros((-B % 4,) + tuple(x.shape[1:]))], dim=0)
        B = x.shape[0]
    if G > 1:
        if units is not None and units[0] > 1:
            raise _lib.BBBHipError("several steps per launch and work units do not combine")
        if B % G or (B // G) % (8 if bf16 else 4):
            raise _lib.BBBHipError("several steps per launch: every batch must hold a multiple of 4 (bf16: 8) images")
        B = B // G
        E = G * draws
        streams = 1
    x_div0, x_off0, nb = (draws if (G > 1 and draws > 1) else 1), 0, G
    if share is not None:
        if G > 1 or (units is not None and units[0] > 1):
            raise _lib.BBBHipError("a share of a group of steps combines with neither work units nor whole groups")
        D, off = int(share[0]), int(share[1])
        nb = -(-(draws + off) // D)
        if not 0 <= off < D or B % nb or (B // nb) % (8 if bf16 else 4):
            raise _lib.BBBHipError("share of a group of steps: x must hold the batches it touches, multiples of 4 (bf16: 8) images")
        B = B // nb
        x_div0, x_off0 = (D, off) if D > 1 else (1, 0)
        streams = 1
    if units is not None and units[0] > 1:
        S, lo, hi = units
        if B % S or (B // S) % (8 if bf16 else 4):
            raise _lib.BBBHipError("work units: batch slices must hold a multiple of 4 (bf16: 8) images")
        E, B = hi - lo, B // S
        j_lo = lo // S
        n_draws = (hi - 1) // S - j_lo + 1                     # weight sets this rank needs
        call0 = call0 + j_lo
        ukw = dict(units=(S, lo % S), n_units=E)
        streams = 1
    else:
        S, n_draws = 1, E
    if bf16 and (lrt or B % 8 != 0):
        raise _lib.BBBHipError("the bf16 path covers BBB (non-LRT) layers and batch sizes that are multiples of 8")
    # split-bf16 mode: BBB layers with Cin % 32 == 0 (never the first one: its input is the caller's fp32 batch) run on the
    # MFMA-ready-operand kernel (ops.conv2d_c8x3_forward) -- their input travels channel-interleaved and already split ("c8 S3"),
    # their weights come tap-major from the parameter pass.  Which kernel a layer takes is a property of the LAYER (not of the
    # launch size), so that a work unit, a share of a group of steps and the whole step are the same bits.
    children = flat_children(net)
    last_bayes = max((i for i, m in enumerate(children) if isinstance(m, (_BBBLayer, _LRTLayer))), default=-1)
    tail_is_last = last_bayes == len(children) - 1
    split_any = (precision == "bf16x3" or ops.current_config().gemm_mode == "bf16x3") and not bf16 and tail_is_last
    split_mode = split_any and not lrt and bool(bbb)
    # ... LRT models (every Bayesian layer local-reparameterisation) the same way, on the kernel's LRT form: six-plane slabs (values +
    # squares), W_mu / W_sigma^2 tap-major (rearranged once per launch: LRT weights do not depend on the draw)
    lrt_mode = split_any and bool(lrt) and not bbb and ops.current_config().c8x3
    c8_set = set()
    if lrt_mode:
        for l in lrt[1:]:
            cin = l.in_channels if isinstance(l, _LRTConv) else l.in_features
            cout = l.out_channels if isinstance(l, _LRTConv) else l.out_features
            if ops.c8x3_layer_ok(cin, cout, is_logits=(l is children[last_bayes])):
                c8_set.add(l)
    if split_mode and ops.current_config().c8x3:
        for l in bbb[1:]:
            cin = l.in_channels if isinstance(l, _BBBConv) else l.in_features
            cout = l.out_channels if isinstance(l, _BBBConv) else l.out_features
            if ops.c8x3_layer_ok(cin, cout, is_logits=(l is children[last_bayes])):
                c8_set.add(l)
    # ... and a strided first layer on few channels (AlexNet conv1) joins the chain in space-to-depth form (ops.s2d_layer_ok): its block
    # image is cut from the caller's NCHW batch in c8 S3 directly, its dense weight draws are rearranged by one small launch
    s2d_first = None
    first_l = bbb[0] if (split_mode and bbb) else (lrt[0] if (lrt_mode and lrt) else None)
    if first_l is not None and ops.current_config().c8x3 and ops.current_config().c8x3_s2d and children and children[0] is first_l and \
            isinstance(first_l, (_BBBConv, _LRTConv)) and first_l not in c8_set and x.dim() == 4 and last_bayes != 0:
        l0 = first_l
        # (an LRT first layer whose input AND weights are the same for all E > 1 draws stays on the fp32 kernel: it runs its two
        # contractions ONCE and samples E times -- the block form would run them E times: 6.1 against 5.1 M samples/s at bs 512 x 10)
        # -- for EVERY partition of such a step (work units, shares of a group): which kernel a layer takes is a property of the step)
        lrt_shared_first = lrt_mode and int(draws) > 1
        if not lrt_shared_first and \
                ops.s2d_layer_ok(l0.in_channels, l0.out_channels, l0.kernel_size, l0.stride, l0.padding, l0.dilation, x.shape[2], x.shape[3]):
            s2d_first = l0
    if bbb:
        sampled, kl = _sample_all_bf16(bbb, n_draws, seed, call0, timers) if bf16 else _sample_all(bbb, n_draws, seed, call0, timers, tm=c8_set)
    if lrt:
        variances, k2 = _variances_all(lrt, timers)
        kl = k2 if kl is None else kl + k2
    to_cb = ops.to_batch_innermost_bf16 if bf16 else ops.to_batch_innermost
    nblk = (S if S > 1 else nb) if (S > 1 or G > 1 or share is not None) else 1
    xs2d = w_s2d = None
    lrt_tm = {}                                                   # LRT layer -> (W_mu, W_sigma^2) tap-major [Cout, taps, Cin]
    for l in (c8_set if lrt_mode else ()):
        wm = l.W_mu.detach()
        wv = variances[l][0]
        if wm.dim() == 2:
            lrt_tm[l] = (wm.reshape(wm.shape[0], 1, wm.shape[1]), wv.reshape(wv.shape[0], 1, wv.shape[1]))
        else:
            lrt_tm[l] = (_run(timers, "layout", None, lambda wm=wm: ops.w_tap_major(wm.unsqueeze(0))[0]),
                         _run(timers, "layout", None, lambda wv=wv: ops.w_tap_major(wv.unsqueeze(0))[0]))
    if s2d_first is not None:
        l0 = s2d_first
        xt = None
        xs2d = _run(timers, "layout", None, lambda: ops.s2d_c8s3(x, nblk, l0.kernel_size, l0.stride, l0.padding, squares=lrt_mode))   # [nblk, 3|6, C'/8, Hb, Wb, B, 8]
        if lrt_mode:
            w_s2d = (_run(timers, "layout", None, lambda: ops.w_s2d_tap_major(l0.W_mu.detach().unsqueeze(0), l0.stride)[0]),
                     _run(timers, "layout", None, lambda: ops.w_s2d_tap_major(variances[l0][0].unsqueeze(0), l0.stride)[0]))
        else:
            w_s2d = _run(timers, "layout", None, lambda: ops.w_s2d_tap_major(sampled[l0][0], l0.stride))              # [n_draws, Cout, m*m, C']
    elif nblk > 1 or S > 1 or G > 1 or share is not None:       # [S, C, H, W, B/S]: one batch-innermost block per slice / per step
        xt = ops.to_batch_innermost_bf16_slices(x, nblk) if bf16 else ops.to_batch_innermost_slices(x, nblk)
    else:
        xt = to_cb(x).unsqueeze(0)                              # [1, C, H, W, B], shared by all draws
    n_out = getattr(children[last_bayes], "out_features", None) if tail_is_last else None
    logits_buf = torch.empty((E, n_out, B), dtype=torch.float32, device=x.device) if n_out is not None else None

    bf16x3 = True if precision == "bf16x3" else None        # None: the current LaunchConfig's gemm_mode decides
    # split-bf16 mode, steps large enough that every conv launch takes that kernel: the activations between the layers travel in
    # the split format S3 (three bf16 planes holding the exact fp32 values; ops.conv2d_chwn_forward x_s3 / out_s3) -- each
    # element is cut into its pieces ONCE, by the launch that produces it, instead of by every workgroup that stages it
    s3_chain = split_mode and not c8_set and B % 8 == 0 and E * B >= ops.current_config().s3_min_images

    def run(e0, e1):
        """Layers for draws [e0, e1) on the current stream -> logits [e1-e0, C, B] (or None: fall back)."""
        nonlocal logits_buf
        B = xs2d.shape[5] if xs2d is not None else xt.shape[-1]
        Es = e1 - e0
        h = xt
        s3 = False                     # h is an S3 tensor [E, 3, C, H, W, B] (split-bf16 chain)
        c8s3 = False                   # h is a c8 S3 tensor [E, 3, C / 8, H, W, B, 8] (split-bf16 chain over MFMA-ready operands)
        boff = int(b_offset)           # global index of the first local "image" (rows multiply at a flatten that cuts images up)
        per_slice = bool(ukw)          # work units: until the first Bayesian layer, h is one block per batch slice
        x_div, x_off = x_div0, x_off0  # several steps per launch: the first layer's slab e reads batch (e + x_off) // x_div
        i = 0
        while i < len(children):
            mod = children[i]
            nxt = children[i + 1] if i + 1 < len(children) else None
            act = _act_name(nxt) if nxt is not None else None
            if mod is s2d_first and i == 0 and lrt_mode:
                # an LRT first layer in space-to-depth form (values and squares of the block image; no pooled form: the pool follows)
                wm_, wv_ = w_s2d
                m_ = int(round(wm_.shape[1] ** 0.5))
                zb = ops.s2d_zero_border(mod.in_channels, mod.kernel_size, mod.stride, mod.padding, x.shape[2], x.shape[3])
                ukw3 = dict(ukw, x_per_slice=True) if ukw else ({"x_div": x_div, "x_off": x_off, "n_slabs": Es} if x_div > 1 else {"n_slabs": Es})
                x_div = 1
                per_slice = False
                fl = conv_flops(B, mod.in_channels, x.shape[2], x.shape[3], wm_.shape[0], *mod.kernel_size, mod.stride, mod.padding, mod.dilation, Es, 2) \
                    if timers is not None else None
                h = _run(timers, "lrt_gemm", fl, lambda wm_=wm_, wv_=wv_, m_=m_, act=act, ukw3=ukw3, zb=zb, mod=mod, boff=boff:
                         ops.lrt_conv2d_c8x3_forward(xs2d, wm_, wv_, mod.bias_mu if mod.use_bias else None, variances[mod][1], (m_, m_), seed,
                                                     call0 + e0, mod._stream_base + 2, 1, 0, 1, act=act, b_offset=boff, zero_border=zb, **ukw3))
                c8s3 = True
                i += (1 if act is not None else 0) + 1
                continue
            if mod is s2d_first and i == 0:
                # the first layer in space-to-depth form: an m x m layer, stride 1, no padding, on the block image; [activation ->]
                # MaxPool2d(2, 2) inside the launch (every window walks the same taps: the parallel-window form)
                w, b = w_s2d, sampled[mod][1]
                if not ukw:
                    w = w[e0:e1]
                    b = None if b is None else b[e0:e1]
                m_ = int(round(w.shape[2] ** 0.5))
                g_ = ops.s2d_geometry(mod.in_channels, mod.kernel_size, mod.stride, mod.padding, mod.dilation, x.shape[2], x.shape[3])
                pool_at = i + (2 if act is not None else 1)
                pool_mod = children[pool_at] if pool_at < len(children) and isinstance(children[pool_at], nn.MaxPool2d) else None
                fuse_pool = pool_mod is not None and g_[4] % 2 == 0 and g_[5] % 2 == 0 and ops.is_pool_2x2(pool_mod)
                ukw3 = dict(ukw, x_per_slice=True) if ukw else ({"x_div": x_div, "x_off": x_off} if x_div > 1 else {})
                x_div = 1
                per_slice = False
                fl = conv_flops(B, mod.in_channels, x.shape[2], x.shape[3], w.shape[1], *mod.kernel_size, mod.stride, mod.padding, mod.dilation, Es) \
                    if timers is not None else None
                zb = ops.s2d_zero_border(mod.in_channels, mod.kernel_size, mod.stride, mod.padding, x.shape[2], x.shape[3])
                h = _run(timers, "conv_gemm", fl, lambda w=w, b=b, m_=m_, act=act, ukw3=ukw3, fuse_pool=fuse_pool, zb=zb:
                         ops.conv2d_c8x3_forward(xs2d, w, b, (m_, m_), 1, 0, 1, act=act, pool=fuse_pool, zero_border=zb, **ukw3))
                c8s3 = True
                i += (1 if act is not None else 0) + (1 if fuse_pool else 0) + 1
                continue
            if isinstance(mod, (_BBBLayer, _LRTLayer)):
                is_conv = isinstance(mod, (_BBBConv, _LRTConv))
                geom = (mod.stride, mod.padding, mod.dilation) if is_conv else (1, 0, 1)
                use_c8 = mod in c8_set
                if c8s3 and not use_c8:
                    h, c8s3 = ops.c8s3_to_f32(h), False
                if use_c8 and not c8s3:
                    hf = h if is_conv else h.reshape(h.shape[0], mod.in_features, 1, 1, -1)
                    if hf.dim() != 5 or hf.shape[-1] != B:
                        return None
                    h = _run(timers, "layout", None, lambda hf=hf: ops.c8s3_from_f32(hf, squares=lrt_mode))
                    c8s3 = True
                if c8s3:
                    h5 = h if is_conv else h.reshape(h.shape[0], h.shape[1], mod.in_features // 8, 1, 1, h.shape[5], 8)
                    if h5.shape[5] != B or h5.shape[2] * 8 != (mod.in_channels if is_conv else mod.in_features):
                        return None
                elif s3:
                    h5 = h if is_conv else h.reshape(h.shape[0], 3, mod.in_features, 1, 1, -1)
                else:
                    h5 = h if is_conv else h.reshape(h.shape[0], mod.in_features, 1, 1, -1)
                c8 = bf16 and h5.dim() == 6                      # channel-interleaved [E, C / 8, H, W, B, 8] (ops.to_c8), written by the layer before
                if not c8s3 and (h5.dim() != (6 if (s3 or c8) else 5) or h5.shape[4 if c8 else -1] != B):
                    return None                                  # flatten quirk etc.: caller falls back
                ukw2 = dict(ukw, x_per_slice=per_slice) if ukw else ({"x_div": x_div, "x_off": x_off} if x_div > 1 else {})
                per_slice = False
                x_div = 1
                if isinstance(mod, _BBBLayer) and bf16:
                    w, b = sampled[mod]
                    if not ukw:
                        w = w[e0:e1]
                        b = None if b is None else b[e0:e1]
                    ckk = (mod.in_channels, *mod.kernel_size) if is_conv else (mod.in_features, 1, 1)
                    fl = conv_flops(B, h5.shape[1] * (8 if c8 else 1), h5.shape[2], h5.shape[3], w.shape[1], ckk[1], ckk[2], *geom, Es) \
                        if timers is not None else None
                    is_logits = logits_buf is not None and i == last_bayes and not is_conv
                    dst = logits_buf[e0:e1] if is_logits else None
                    tapm = is_conv and ops.bf16_tap_major(tuple(mod.W_mu.shape))
                    of32 = i == last_bayes and tail_is_last
                    # [activation ->] MaxPool2d after a first layer with a short contraction: one launch (ops.bf16_pool_fusion_ok)
                    pool_at = i + (2 if act is not None else 1)
                    pool_mod = children[pool_at] if pool_at < len(children) and isinstance(children[pool_at], nn.MaxPool2d) else None
                    fuse_pool = is_conv and ops.bf16_pool_fusion_ok(ckk, tapm, of32, pool_mod, tuple(h5.shape), geom, Es)
                    pool_ks = (pool_mod.kernel_size, pool_mod.stride if pool_mod.stride is not None else pool_mod.kernel_size) if fuse_pool else None
                    # the pooled first layer writes its output channel-interleaved when the layer that reads it has the strip form over
                    # that layout (3Conv3FC conv1 + pool1 -> conv2; ops.bf16_c8_input_ok): same values, 8 channels of an image adjacent
                    out_c8 = False
                    nb = children[pool_at + 1] if (fuse_pool and pool_at + 1 < len(children)) else None
                    if isinstance(nb, _BBBConv) and w.shape[1] % 8 == 0 and dst is None:
                        hw = [(h5.shape[2 + a] + 2 * _p2(geom[1])[a] - _p2(geom[2])[a] * (ckk[1 + a] - 1) - 1) // _p2(geom[0])[a] + 1 for a in (0, 1)]
                        hw = [(v - _p2(pool_ks[0])[a]) // _p2(pool_ks[1])[a] + 1 for a, v in enumerate(hw)]
                        out_c8 = ops.bf16_c8_input_ok((nb.in_channels, *nb.kernel_size), (nb.stride, nb.padding, nb.dilation),
                                                      ops.bf16_tap_major(tuple(nb.W_mu.shape)), pool_at + 1 == last_bayes and tail_is_last,
                                                      (hw[0], hw[1], B), Es)
                    # (operands bound as defaults: bench.py's LaunchRecorder replays these closures after the loop has moved on)
                    y = _run(timers, "conv_gemm", fl,
                             lambda h5=h5, w=w, b=b, ckk=ckk, geom=geom, act=act, dst=dst, ukw2=ukw2, tapm=tapm, of32=of32, pool_ks=pool_ks,
                             out_c8=out_c8:
                             ops.conv2d_chwn_bf16_forward(h5, w, b, ckk, *geom, act=act, out_f32=of32, out=dst, tap_major=tapm,
                                                          pool=pool_ks, out_c8=out_c8, **ukw2))
                    if fuse_pool:
                        i += 1                                   # the pooling module is done too
                elif use_c8 and lrt_mode:
                    wm_, wv_ = lrt_tm[mod]
                    ks = mod.kernel_size if is_conv else (1, 1)
                    fl = conv_flops(B, h5.shape[2] * 8, h5.shape[3], h5.shape[4], wm_.shape[0], ks[0], ks[1], *geom, Es, 2) if timers is not None else None
                    is_logits = logits_buf is not None and i == last_bayes and not is_conv
                    of32 = i == last_bayes
                    dst = logits_buf[e0:e1] if is_logits else None
                    ukw3 = {k: v for k, v in ukw2.items() if k != "x_per_slice"}
                    if not ukw:
                        ukw3["n_slabs"] = Es
                    y = _run(timers, "lrt_gemm", fl, lambda h5=h5, wm_=wm_, wv_=wv_, ks=ks, geom=geom, act=act, dst=dst, ukw3=ukw3, of32=of32, mod=mod, boff=boff:
                             ops.lrt_conv2d_c8x3_forward(h5, wm_, wv_, mod.bias_mu if mod.use_bias else None, variances[mod][1], ks, seed,
                                                         call0 + e0, mod._stream_base + 2, *geom, act=act, out_f32=of32, out=dst, b_offset=boff,
                                                         **ukw3))
                    c8s3 = not of32
                elif use_c8:
                    w, b = sampled[mod]
                    if not ukw:
                        w = w[e0:e1]
                        b = None if b is None else b[e0:e1]
                    ks = mod.kernel_size if is_conv else (1, 1)
                    w = w.reshape(w.shape[0], w.shape[1], ks[0] * ks[1], -1)      # tap-major rows (a linear layer's rows as they are)
                    fl = conv_flops(B, h5.shape[2] * 8, h5.shape[3], h5.shape[4], w.shape[1], ks[0], ks[1], *geom, Es) if timers is not None else None
                    is_logits = logits_buf is not None and i == last_bayes and not is_conv
                    of32 = i == last_bayes
                    dst = logits_buf[e0:e1] if is_logits else None
                    ukw3 = {k: v for k, v in ukw2.items() if k != "x_per_slice"}
                    y = _run(timers, "conv_gemm", fl, lambda h5=h5, w=w, b=b, ks=ks, geom=geom, act=act, dst=dst, ukw3=ukw3, of32=of32:
                             ops.conv2d_c8x3_forward(h5, w, b, ks, *geom, act=act, out_f32=of32, out=dst, **ukw3))
                    c8s3 = not of32
                elif isinstance(mod, _BBBLayer):
                    w, b = sampled[mod]
                    if not ukw:
                        w = w[e0:e1]
                        b = None if b is None else b[e0:e1]
                    if not is_conv:
                        w = w.reshape(w.shape[0], mod.out_features, mod.in_features, 1, 1)
                    fl = conv_flops(B, *h5.shape[-4:-1], w.shape[1], w.shape[3], w.shape[4], *geom, Es) if timers is not None else None
                    dst = logits_buf[e0:e1] if (logits_buf is not None and i == last_bayes and not is_conv) else None
                    o_s3 = s3_chain and i != last_bayes          # intermediate layers hand their output on already split
                    # [activation ->] MaxPool2d(2, 2) after a conv layer: one launch with it when the launch is large enough
                    # (ops.pool_fusion_ok; same bits as the separate pooling launch)
                    pool_at = i + (2 if act is not None else 1)
                    pool_mod = children[pool_at] if pool_at < len(children) and isinstance(children[pool_at], nn.MaxPool2d) else None
                    # (a chain over MFMA-ready operands: the layers outside it -- the first one -- take the fp32 kernel, pooled form included)
                    bx3 = False if c8_set else bf16x3
                    fuse_pool = (pool_mod is not None and is_conv and not s3 and not o_s3 and (bf16x3 is None or bool(c8_set)) and
                                 ops.pool_fusion_ok(tuple(h5.shape), tuple(w.shape), *geom, Es, pool_mod, fp32_kernel=bool(c8_set)))
                    y = _run(timers, "conv_gemm", fl, lambda h5=h5, w=w, b=b, geom=geom, act=act, dst=dst, ukw2=ukw2, s3=s3, o_s3=o_s3, fuse_pool=fuse_pool, bx3=bx3:
                             ops.conv2d_chwn_forward(h5, w, b, *geom, act=act, out=dst, bf16x3=bx3, x_s3=s3, out_s3=o_s3,
                                                     pool=fuse_pool, **ukw2))
                    s3 = o_s3
                    if fuse_pool:
                        i += 1                                   # the pooling module is done too
                else:
                    w_var, b_var = variances[mod]
                    w_mu = mod.W_mu
                    if not is_conv:
                        shp = (mod.out_features, mod.in_features, 1, 1)
                        w_mu, w_var = w_mu.reshape(shp), w_var.reshape(shp)
                    shared_in = h5.shape[0] == 1 and Es > 1 and not ukw and G == 1 and share is None
                    fl = conv_flops(B, h5.shape[1], h5.shape[2], h5.shape[3], w_mu.shape[0], w_mu.shape[2], w_mu.shape[3],
                                    *geom, 1 if shared_in else Es, 2) if timers is not None else None
                    if shared_in:
                        # same input AND same (mu, sigma^2) weights for every draw: the two contractions run once, the E
                        # draws differ only in the epilogue noise (bitwise the same result as E full launches)
                        _, am, av = _run(timers, "lrt_gemm", fl,
                                         lambda h5=h5, w_mu=w_mu, w_var=w_var, b_var=b_var, mod=mod, geom=geom:
                                         ops.lrt_conv2d_chwn_forward(h5, w_mu, w_var, mod.bias_mu if mod.use_bias else None,
                                                                     b_var, seed, call0 + e0, mod._stream_base + 2, *geom,
                                                                     sample=False, want_moments=True, act=None))
                        y = _run(timers, "lrt_sample", None,
                                 lambda am=am, av=av, mod=mod, act=act, boff=boff: ops.lrt_sample_chwn(am, av, Es, seed, call0 + e0, mod._stream_base + 2, act=act, b_offset=boff))
                    else:
                        y = _run(timers, "lrt_gemm", fl,
                                 lambda h5=h5, w_mu=w_mu, w_var=w_var, b_var=b_var, mod=mod, geom=geom, act=act, ukw2=ukw2, boff=boff:
                                 ops.lrt_conv2d_chwn_forward(h5, w_mu, w_var, mod.bias_mu if mod.use_bias else None, b_var,
                                                             seed, call0 + e0, mod._stream_base + 2, *geom, sample=True,
                                                             act=act, b_offset=boff, **ukw2,
                                                             **({"n_slabs": Es} if "x_div" in ukw2 else {}))[0])
                h = y
                if act is not None:
                    i += 1
            elif isinstance(mod, FlattenLayer):
                if c8s3:
                    if h.shape[3] * h.shape[4] == 1 and h.shape[2] * 8 == mod.num_features:
                        i += 1                                   # [E, 3, F / 8, 1, 1, B, 8] already is the flattened feature order
                        continue
                    h, c8s3 = ops.c8s3_to_f32(h), False
                if s3 and h.shape[2] * h.shape[3] * h.shape[4] != mod.num_features:
                    h, s3 = ops.s3_to_f32(h), False              # the flatten quirk below works on the fp32 tensor
                if h.dim() != (6 if s3 else 5):
                    return None
                chw = h.shape[-4] * h.shape[-3] * h.shape[-2]
                if s3:
                    h = h.reshape(h.shape[0], 3, mod.num_features, 1, 1, B)
                elif chw == mod.num_features:
                    h = h.reshape(h.shape[0], mod.num_features, 1, 1, B)
                else:
                    # the reference's view(-1, num_features) on a larger map (AlexNet on 224x224: [B,128,7,7] -> [B*49,128])
                    # cuts each image's NCHW memory into rows of num_features: go through the NCHW order once, on this
                    # small tensor, and continue batch-innermost with B' = B*chw/num_features "images"
                    if chw % mod.num_features != 0 or bf16:
                        return None
                    rows = h.permute(0, 4, 1, 2, 3).reshape(h.shape[0], -1, mod.num_features)     # [E|1, B', F]
                    boff *= chw // mod.num_features
                    B = rows.shape[1]
                    if B % 4 != 0:
                        return None
                    h = rows.permute(0, 2, 1).contiguous().reshape(h.shape[0], mod.num_features, 1, 1, B)
                    if logits_buf is not None and logits_buf.shape[2] != B:
                        logits_buf = torch.empty((E, n_out, B), dtype=torch.float32, device=x.device)
            elif isinstance(mod, nn.MaxPool2d):
                pool = ops.maxpool_c8s3 if c8s3 else ops.maxpool_chwn_s3 if s3 else (ops.maxpool_chwn_bf16 if bf16 else ops.maxpool_chwn)
                h = _run(timers, "maxpool", None, lambda: pool(h, mod.kernel_size, mod.stride))
            elif isinstance(mod, nn.ReLU):
                if s3:
                    h, s3 = ops.s3_to_f32(h), False
                if c8s3:
                    h, c8s3 = ops.c8s3_to_f32(h), False
                h = torch.relu(h)
            else:
                if s3:
                    h, s3 = ops.s3_to_f32(h), False
                if c8s3:
                    h, c8s3 = ops.c8s3_to_f32(h), False
                h = F.softplus(h)
            i += 1
        if s3:
            h, s3 = ops.s3_to_f32(h), False
        if c8s3:
            h, c8s3 = ops.c8s3_to_f32(h), False
        if h.shape[0] == 1 and Es > 1:
            h = h.expand(Es, *h.shape[1:])
        h = h.reshape(Es, -1, B)
        if logits_buf is not None and h.shape[1] == logits_buf.shape[1]:
            dst = logits_buf[e0:e1]
            if h.data_ptr() != dst.data_ptr():
                dst.copy_(h)
            return dst
        return h

    nsplit = max(1, min(int(streams), E))
    if nsplit == 1 or timers is not None:
        stats["launch"] = "layers"
        out = run(0, E)
        if out is None:
            return None
    else:
        stats["launch"] = "layers"
        # Independent sub-ensembles on separate HIP streams: each GEMM launch is one wave of workgroups with a ramp
        # and a tail, and the other stream's kernels fill those bubbles (and the launch gaps).
        main = torch.cuda.current_stream(x.device)
        pool = _side_streams(x.device, nsplit)
        bounds = [draw_range(E, r, nsplit) for r in range(nsplit)]
        parts = []
        for st, (e0, e1) in zip(pool, bounds):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                parts.append(run(e0, e1))
        for st in pool:
            main.wait_stream(st)
        if any(pt is None for pt in parts):
            return None
        out = logits_buf if (logits_buf is not None and all(pt.data_ptr() == logits_buf[b0:b1].data_ptr()
                                                            for pt, (b0, b1) in zip(parts, bounds))) \
            else torch.cat(parts, dim=0)
    stats["path"] = "chwn"
    if rows_real is not None:
        out = out[:, :, :rows_real]                              # (a view: the consumers' .contiguous() compacts it)
    return out, kl                                               # logits stay batch-innermost: [E, C, B]


_stream_pool = {}


def _side_streams(device, n):
    key = (torch.device(device).index, n)
    if key not in _stream_pool:
        _stream_pool[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _stream_pool[key]


_lane_pool = {}


def _lane_streams(device, n):
    """The first n streams of the device's lane pool (created once, in order, so that lanes 0..n-1 always sit on the same
    hardware queues)."""
    key = torch.device(device).index
    pool = _lane_pool.setdefault(key, [])
    while len(pool) < max(n, 4):                         # created AND first used back to back
        st = torch.cuda.Stream(device=device)
        # a HIP stream gets its hardware queue at its first launch, and queues are spread over the 4 compute pipes in creation
        # order: four lanes that first ran back to back sit on four pipes; a lane first used after other streams had their first
        # launch can share a pipe with another lane (measured: the same 4-lane pipeline 0.057 or 0.100 ms per step)
        with torch.cuda.stream(st):
            torch.zeros(1, device=device)
        pool.append(st)
    return pool[:n]


def _loop_logits(net, x, draws, seed, call0, eps=None):
    """The reference's own loop, `for j in range(E): net(x)` under call indices call0 + j, for models the batched paths
    cannot flatten (a Bayesian layer inside a module with its own forward): same numbers, E times the launches."""
    if eps is not None:
        raise _lib.BBBHipError("external eps needs a model made of direct-child / Sequential-nested Bayesian layers")
    stats["path"] = "loop"
    saved = dict(rng._state)
    try:
        rng.manual_seed(seed, call=call0)            # pinned for the loop: torch's generator is not touched
        outs, kl = [], None
        for _ in range(draws):
            y, k = net(x)
            outs.append(y)
            kl = k
    finally:
        rng._state.clear()
        rng._state.update(saved)
    return torch.stack(outs), kl


def mc_logits(net, x, draws, seed, call0, fuse_act=True, timers=None, eps=None, layout="auto", streams=1, precision="fp32"):
    """E stochastic forwards of `net` on the same batch x -> (logits [E, B', C], kl of ONE forward).

    Equivalent to `[net(x)[0] for _ in range(E)]` under noise calls call0 .. call0+E-1 (B' = B except for
    the 224x224 AlexNet flatten quirk, where B' = B*49).  layout "auto" takes the batch-innermost fast path
    when it applies (inference, B % 4 == 0, known module kinds), "nchw" forces the reference layout."""
    _lib.require_device(x)
    _check_precision(precision, net, x, layout != "nchw" and eps is None and fuse_act)
    if layout != "nchw" and eps is None and fuse_act and _chwn_ok(net, x):
        out = _mc_logits_chwn(net, x, draws, seed, call0, timers, streams, precision)
        if out is not None:
            return out[0].permute(0, 2, 1).contiguous(), out[1]      # API layout [E, B, C]
    if precision == "bf16":
        raise _lib.BBBHipError("this model / input does not fit the batch-innermost path, which is the only bf16 path")
    if flat_children(net) is None:
        return _loop_logits(net, x, draws, seed, call0, eps)
    stats["path"] = "nchw"
    layers = bayesian_layers(net)
    bbb = [l for l in layers if isinstance(l, _BBBLayer)]
    lrt = [l for l in layers if isinstance(l, _LRTLayer)]
    kl = None
    sampled, variances = {}, {}
    if bbb:
        sampled, k1 = _sample_all(bbb, draws, seed, call0, timers, eps)
        kl = k1
    if lrt:
        variances, k2 = _variances_all(lrt, timers)
        kl = k2 if kl is None else kl + k2
    E = draws
    h = x.unsqueeze(0)                      # [1, B, ...] shared by all draws until a Bayesian layer splits it
    children = flat_children(net)
    i = 0
    while i < len(children):
        mod = children[i]
        nxt = children[i + 1] if i + 1 < len(children) else None
        act = _act_name(nxt) if (fuse_act and nxt is not None) else None
        if isinstance(mod, (_BBBLayer, _LRTLayer)):
            is_conv = isinstance(mod, (_BBBConv, _LRTConv))
            if is_conv:
                h5 = h
                geom = (mod.stride, mod.padding, mod.dilation)
            else:
                h5 = h.reshape(h.shape[0], -1, mod.in_features, 1, 1)
                geom = (1, 0, 1)
            if isinstance(mod, _BBBLayer):
                w, b = sampled[mod]
                if not is_conv:
                    w = w.reshape(E, mod.out_features, mod.in_features, 1, 1)
                kdim = w[0][0].numel()                          # Cin*kh*kw
                flops = lambda yy: 2.0 * yy.numel() * kdim      # 2*M*N*K summed over draws
                if torch.is_grad_enabled() and (h5.requires_grad or w.requires_grad):
                    y = ops.conv2d(h5, w, b, *geom)
                    if act is not None:
                        y = F.relu(y) if act == "relu" else F.softplus(y)
                else:
                    y = _run(timers, "conv_gemm", flops, lambda: ops.conv2d_forward(h5, w, b, *geom, act=act))
            else:
                w_var, b_var = variances[mod]
                w_mu = mod.W_mu
                if not is_conv:
                    shp = (mod.out_features, mod.in_features, 1, 1)
                    w_mu, w_var = w_mu.reshape(shp), w_var.reshape(shp)
                if h5.shape[0] == 1 and E > 1:
                    h5 = h5.expand(E, *h5.shape[1:])
                if torch.is_grad_enabled() and (h5.requires_grad or w_mu.requires_grad):
                    y = ops.lrt_conv2d(h5, w_mu, w_var, mod.bias_mu if mod.use_bias else None, b_var, seed, call0,
                                       mod._stream_base + 2, *geom, sample=True)
                    if act is not None:
                        y = F.relu(y) if act == "relu" else F.softplus(y)
                else:
                    kdim = w_mu[0].numel()
                    flops = lambda yy: 4.0 * yy.numel() * kdim   # two contractions
                    y = _run(timers, "lrt_gemm", flops,
                             lambda: ops.lrt_conv2d_forward(h5, w_mu, w_var, mod.bias_mu if mod.use_bias else None, b_var,
                                                            seed, call0, mod._stream_base + 2, *geom, sample=True, act=act)[0])
            if not is_conv:
                y = y.reshape(E, -1, mod.out_features)
            h = y
            if act is not None:
                i += 1                       # the activation module was fused into the epilogue
        elif isinstance(mod, FlattenLayer):
            h = h.reshape(h.shape[0], -1, mod.num_features)
        else:                                # stock torch module (pool, activation, ...): fold draws into batch
            lead = h.shape[:2]
            y = mod(h.reshape(-1, *h.shape[2:]))
            h = y.reshape(*lead, *y.shape[1:])
        i += 1
    if h.shape[0] == 1 and E > 1:
        h = h.expand(E, *h.shape[1:])
    return h, kl


def _local_lse(net, x, draws, seed, call0, mean_over, fuse_act=True, timers=None, streams=1, precision="fp32", units=None, b_offset=0,
               step_end=None, groups=1, param_alias=None, share=None, elbo=None):
    """(log-sum-exp over the local draws of the per-draw log_softmax [B, C], kl of one forward), staying in the
    batch-innermost layout end to end when the fast path applies.  units = (S, lo, hi): the local work is the unit range
    lo..hi-1 instead of `draws` whole draws (call0 = call index of draw 0); rows of slices without a local unit are -inf.
    step_end = (scale, counter, counter_add) (a captured step, GraphedMC): the tail launch also writes kl * scale and advances the
    device-side call counter; then the second return value is the SCALED kl, and step_end[3] is set to True -- on the paths without
    the fused tail the caller does both itself.
    groups = G > 1: x holds G batches and the launches run G consecutive steps of `draws` forwards each (_mc_logits_chwn);
    -> [G * B, C], block g = step g's result.
    share = (D, off): one rank's part of a group of steps (see _mc_logits_chwn; `draws` = its slabs, call0 = its first call)
    -> [n * B, C] for the n = ceil((draws + off) / D) steps it touches: block k = log-sum-exp over ITS draws of that step, no
    mean (group_share / GraphedMC steps > 1 with a process group combine the ranks' blocks).
    elbo = [target, beta, train_size, None] (a training step, train.train_step): on the batch-innermost autograd path the loss
    nll_loss(lse, target) * train_size + beta * kl is formed by the tail's own launches (ops.elbo_cb_autograd) and left in elbo[3];
    elbo[3] still None on return: the caller forms it."""
    _check_precision(precision, net, x, fuse_act)
    if share is not None:
        D, off = int(share[0]), int(share[1])
        n_steps = -(-(draws + off) // D)
        out = _mc_logits_chwn(net, x, draws, seed, call0, timers, 1, precision, share=(D, off))
        if out is None:
            raise _lib.BBBHipError("a share of a group of steps needs the batch-innermost path")
        if step_end is not None and timers is None:
            lse, klf = ops.mc_tail_share(out[0], n_steps, D, off, step_end=(out[1], step_end[0], step_end[1], step_end[2]))
            step_end[3] = True
            return lse, klf
        lse = _run(timers, "mc_tail", 0, lambda: ops.mc_tail_share(out[0], n_steps, D, off))
        return lse, out[1]
    if int(groups) > 1:
        out = _mc_logits_chwn(net, x, draws, seed, call0, timers, 1, precision, groups=groups)
        if out is None:
            raise _lib.BBBHipError("several steps per launch need the batch-innermost path")
        if step_end is not None and timers is None:
            lse, klf = ops.mc_tail_groups(out[0], groups, draws, mean_over=mean_over,
                                          step_end=(out[1], step_end[0], step_end[1], step_end[2]))
            step_end[3] = True
            return lse, klf
        lse = _run(timers, "mc_tail", 0, lambda: ops.mc_tail_groups(out[0], groups, draws, mean_over=mean_over))
        return lse, out[1]
    if units is not None and units[0] > 1:
        out = _mc_logits_chwn(net, x, draws, seed, call0, timers, 1, precision, units=units)
        if out is None:
            raise _lib.BBBHipError("work units need the batch-innermost path (checked by units_ok before planning)")
        if step_end is not None and timers is None:
            lse, klf = ops.mc_tail_units(out[0], units[0], units[1], mean_over=mean_over, step_end=(out[1], step_end[0], step_end[1], step_end[2]))
            step_end[3] = True
            return lse, klf
        lse = _run(timers, "mc_tail", 0, lambda: ops.mc_tail_units(out[0], units[0], units[1], mean_over=mean_over))
        return lse, out[1]
    if fuse_act and _chwn_ok(net, x):
        out = _mc_logits_chwn(net, x, draws, seed, call0, timers, streams, precision, b_offset=b_offset)
        if out is not None:
            if step_end is not None and timers is None:
                lse, klf = ops.mc_tail_cb(out[0], mean_over=mean_over, step_end=(out[1], step_end[0], step_end[1], step_end[2]))
                step_end[3] = True
                return lse, klf
            lse = _run(timers, "mc_tail", 0, lambda: ops.mc_tail_cb(out[0], mean_over=mean_over))
            return lse, out[1]
    if b_offset:
        raise _lib.BBBHipError("a batch offset needs the batch-innermost inference path (checked: this model / input does not fit it)")
    if precision == "bf16":
        raise _lib.BBBHipError("this model / input does not fit the batch-innermost path, which is the only bf16 path")
    if fuse_act and fast_autograd and torch.is_grad_enabled() and timers is None:
        from . import fast_train
        if fast_train.train_path_ok(net, x):
            # training on the batch-innermost kernels: one autograd node for the whole batched forward
            logits_cb, kl1 = fast_train.mc_logits_autograd(net, x, draws, seed, call0, alias=param_alias)
            stats["path"] = "chwn-autograd"
            if elbo is not None and hip_loss_tail[0] and ops.elbo_cb_ok(logits_cb, kl1, elbo[0], elbo[1]):
                elbo[3], lse = ops.elbo_cb_autograd(logits_cb, kl1, elbo[0], elbo[1], elbo[2], mean_over)
                return lse, kl1
            if logits_cb.shape[0] <= 512 and hip_loss_tail[0]:
                lse = ops.mc_tail_cb_autograd(logits_cb, mean_over)       # log_softmax + logmeanexp, forward and backward one launch each
            else:
                lse = torch.logsumexp(F.log_softmax(logits_cb.permute(0, 2, 1), dim=2), dim=0) \
                    - (math.log(mean_over) if mean_over > 0 else 0.0)
            return lse, kl1
    logits, kl1 = mc_logits(net, x, draws, seed, call0, fuse_act=fuse_act, timers=timers, layout="nchw")
    if torch.is_grad_enabled() and logits.requires_grad:
        # training extension: differentiable tail through torch ops
        lse = torch.logsumexp(F.log_softmax(logits, dim=2), dim=0) - (math.log(mean_over) if mean_over > 0 else 0.0)
    else:
        lse = _run(timers, "mc_tail", 0, lambda: ops.mc_tail(logits, mean_over=mean_over))
    return lse, kl1


hip_loss_tail = [True]      # training: the loss tail's log_softmax + logmeanexp on the HIP kernels (False: torch ops, as until round 5)


def group_share(num_ens, steps, rank, world):
    """A GROUP of `steps` consecutive Monte-Carlo steps dealt out to `world` ranks: the steps * num_ens draws of the group, in
    draw-major (step, draw) order, are cut into `world` contiguous ranges (draw_range) -- whole draws on whole batches, so every
    rank's launches are as large as a single device's would be for that many draws; a step whose draws straddle a boundary is
    combined by the group's one all_gather.  -> (lo, hi, first local step, number of local steps, draws of the first local step
    that belong to earlier ranks); hi == lo: more ranks than draws."""
    lo, hi = draw_range(int(steps) * int(num_ens), rank, world)
    if hi <= lo:
        return lo, hi, 0, 0, 0
    g_lo, g_hi = lo // num_ens, (hi - 1) // num_ens
    return lo, hi, g_lo, g_hi - g_lo + 1, lo % num_ens


def units_ok(net, x, fuse_act=True):
    """Work-unit sharding runs on the batch-innermost path only, and needs the flatten (if any) to keep one row per image."""
    return bool(fuse_act) and _chwn_ok(net, x, any_batch=False) and output_rows(net, tuple(x.shape)) == x.shape[0]


def shard_plan(net, x, num_ens, rank, world, fuse_act=True, precision="fp32"):
    """How this rank's share of one MC step is cut: -> (S, lo, hi).  S = 1: lo..hi are whole draws (the fallback when the
    fast path does not apply).  Every rank computes the same S from the same shapes."""
    S = 1
    if world > 1 and units_ok(net, x, fuse_act):
        S = plan_slices(num_ens, world, x.shape[0], multiple=8 if precision == "bf16" else 4)
    lo, hi = unit_range(num_ens, S, rank, world)
    return S, lo, hi


def mc_forward(net, x, num_ens, group=None, fuse_act=True, timers=None, kl_mode="sum", streams=1, precision="fp32"):
    """One Monte-Carlo step: -> (log_outputs [B, C], kl).

    kl_mode "sum": kl summed over the num_ens calls as validate_model does (main_bayesian.py:76-77);
            "mean": divided by num_ens as train_model does (main_bayesian.py:51).
    With a process group the (draw x batch-slice) work units are dealt out to the ranks (shard_plan) and the ranks combine
    through one all_gather; every rank returns identical values, equal to the single-device result."""
    world = 1 if group is None else torch.distributed.get_world_size(group)
    rank = 0 if group is None else torch.distributed.get_rank(group)
    rng.assign_stream_ids(net)                       # stream ids by module order: identical on every rank
    seed, call0 = rng.next_calls(num_ens)            # all ranks advance identically
    S, lo, hi = shard_plan(net, x, num_ens, rank, world, fuse_act, precision)
    if hi > lo:
        if S > 1:
            lse, kl1 = _local_lse(net, x, num_ens, seed, call0, 0, fuse_act=fuse_act, timers=timers, precision=precision,
                                  units=(S, lo, hi))
        else:
            lse, kl1 = _local_lse(net, x, hi - lo, seed, call0 + lo, 0 if world > 1 else num_ens, fuse_act=fuse_act,
                                  timers=timers, streams=streams, precision=precision)
        kl_local = kl1 * (float(hi - lo) / S)
    else:                                            # more ranks than work units
        lse, kl_local = None, None
    if world == 1:
        kl = kl_local if kl_mode == "sum" else kl_local / num_ens
        return lse, kl
    shape = (output_rows(net, tuple(x.shape)), getattr(net, "num_classes", None)) if lse is None else None
    return combine_ranks(lse, kl_local, num_ens, group, kl_mode, shape=shape, device=x.device)


def combine_ranks(lse_local, kl_local, num_ens, group, kl_mode="sum", shape=None, device=None):
    """All-gather [B*C + 1] floats per rank, then log-sum-exp over ranks in rank order.
    lse_local: [B, C] log-sum-exp over this rank's work (rows of -inf where it held no unit of a batch slice; None if the
    rank had no work at all -- then `shape` = (B, C) and `device` say what to contribute)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if lse_local is None:
        if shape is None or shape[1] is None:
            raise _lib.BBBHipError("a rank without work must be given the [B, C] shape")
        dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                 if torch.cuda.is_available() else torch.device("cpu"))
        lse_local = torch.full(shape, -float("inf"), device=dev)
        kl_local = torch.zeros((), device=dev)
    B, C = lse_local.shape
    buf = torch.cat([lse_local.reshape(-1), kl_local.detach().reshape(1).to(lse_local.dtype)])
    flat = torch.empty(world * buf.numel(), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(flat, buf, group=group)       # ONE collective per MC step (RCCL on GPUs)
    gathered = flat.view(world, buf.numel())
    blocks = gathered[:, :-1].reshape(world, B, C)
    log_outputs = torch.logsumexp(blocks, dim=0) - math.log(num_ens)
    kl = gathered[:, -1].sum()
    if kl_mode != "sum":
        kl = kl / num_ens
    return log_outputs, kl


def mc_forward_batch_parallel(net, x_local, num_ens, group=None, gather=False, fuse_act=True, precision="fp32", b_offset=None):
    """Data-parallel inference (BASELINE.json configs[4]: 4096 x 3 x 224 x 224 over 8 GPUs, 512 images each): every rank runs
    ALL num_ens draws on ITS images.  The weight noise is keyed by (seed, call, stream, element), so all ranks sample the
    same weights without any communication, and the outputs are exactly the rows a single device would compute for these
    images; KL is identical everywhere.  No collective unless gather=True (one all_gather of the [B_local', C] blocks).
    LRT layers key their activation noise by the global image index: pass b_offset (default rank * B_local)."""
    world = 1 if group is None else torch.distributed.get_world_size(group)
    rank = 0 if group is None else torch.distributed.get_rank(group)
    rng.assign_stream_ids(net)
    if b_offset is None:
        b_offset = rank * x_local.shape[0]
    if not any(isinstance(l, _LRTLayer) for l in bayesian_layers(net)):
        b_offset = 0                                  # weight noise does not depend on the batch
    seed, call0 = rng.next_calls(num_ens)
    lse, kl1 = _local_lse(net, x_local, num_ens, seed, call0, num_ens, fuse_act=fuse_act, precision=precision, b_offset=int(b_offset))
    kl = kl1 * float(num_ens)
    if gather and world > 1:
        import torch.distributed as dist
        full = torch.empty((world * lse.shape[0], lse.shape[1]), dtype=lse.dtype, device=lse.device)
        dist.all_gather_into_tensor(full, lse.contiguous(), group=group)
        return full, kl
    return lse, kl


def _all_gather(recv, send, group):
    torch.distributed.all_gather_into_tensor(recv, send, group=group)


def _eager_gather(recv, send, group, lane_stream, comm_stream):
    """An EAGER collective between two graph replays of a lane, issued on the lane's own communication stream (ordered with the
    lane stream through events), never on the lane stream itself: torch's NCCL watchdog thread keeps querying the completion
    event of an eager collective for up to its polling interval after the call, and this HIP runtime refuses such a query
    (hipErrorCapturedEvent -> the watchdog terminates the process) once the stream the event was recorded on is being
    CAPTURED -- which a later GraphedMC on the same pooled lane stream would do.  Streams that get captured never carry eager
    collectives."""
    comm_stream.wait_stream(lane_stream)
    with torch.cuda.stream(comm_stream):
        _all_gather(recv, send, group)
    lane_stream.wait_stream(comm_stream)


capture_collectives = True     # a sharded GraphedMC step records its ONE all_gather inside the step's hipGraph when the backend allows
                               # it (RCCL does; probed once per process group): one host call per step instead of three
capture_probe_timeout_s = 120.0  # the probe's collectives run under this wall-clock watchdog (a rank that never arrives must not
                                 # hang the job: the others fall back to the eager protocol and say so); generous: it covers the
                                 # creation of two communicators (seconds each on an 8-GPU node)
_capture_probe = weakref.WeakKeyDictionary()     # process group object -> bool (NOT id(group): ids are reused after destruction)
_lane_groups = weakref.WeakKeyDictionary()       # process group object -> {lane: private communicator}
last_protocol = {"collective": None, "reason": None}   # what the last collective_capture_ok decided and why (bench.py prints it)


def _spans_world(group):
    import torch.distributed as dist
    return group is None or dist.get_world_size(group) == dist.get_world_size()


def lane_group(group, lane):
    """The communicator lane `lane` of a pipeline RECORDS into its graphs: always a PRIVATE one over the same ranks (created
    collectively, once per (group, lane), cached on the group object) -- never the caller's `group` itself, whose eager collectives
    (the caller's own, torch's barrier) are serialised by torch but would be unordered with a recorded one: two collectives of ONE
    communicator must never run concurrently or in different orders on different ranks.  dist.new_group has to be entered by every
    rank of the default group, so recording is only offered for groups that span WORLD (collective_capture_ok checks)."""
    import torch.distributed as dist
    if group is None:
        return group
    lanes = _lane_groups.get(group)
    if lanes is None:
        lanes = _lane_groups[group] = {}
    if lane not in lanes:
        backend = dist.get_backend(group)
        g = lanes[lane] = dist.new_group(ranks=dist.get_process_group_ranks(group), backend=backend)
        if backend == "nccl":
            # the communicator must exist before a collective of it is recorded: one eager call, on the CURRENT stream -- never on
            # a stream that will be captured (see _eager_gather)
            dev = torch.device("cuda", torch.cuda.current_device())
            t = torch.zeros((4,), device=dev)
            dist.all_gather_into_tensor(torch.empty((4 * dist.get_world_size(g),), device=dev), t, group=g)
            torch.cuda.synchronize(dev)
    return lanes[lane]


def force_eager_collectives(group):
    """After a stall of a pipeline whose collectives were recorded into its graphs: from now on this group's pipelines use the
    eager protocol, on FRESH lane streams (the old ones may sit behind a collective kernel that never completes)."""
    _capture_probe[group] = False
    last_protocol.update(collective="eager", reason="forced after a stalled recorded collective")
    _lane_pool.clear()
    _stream_pool.clear()


def _with_watchdog(fn, timeout_s):
    """Run fn() on a helper thread; (True, result) or (False, exception | None when it did not return within timeout_s).  A
    collective that never completes cannot be cancelled -- the helper thread is left behind (daemon) and the caller moves on to the
    eager protocol / reports the stall instead of sitting in the driver's lease forever."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as exc:                 # noqa: BLE001 (reported to the caller)
            box["error"] = exc

    dev = torch.cuda.current_device() if torch.cuda.is_available() else None

    def entry():
        if dev is not None:
            torch.cuda.set_device(dev)
        run()

    t = threading.Thread(target=entry, daemon=True, name="bbb-collective-probe")
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        return False, None
    if "error" in box:
        return False, box["error"]
    return True, box.get("value")


def _local_capture_preconditions(group, device):
    """This rank's own view: may its all_gather be recorded?  (reason string when not.)  No collective is entered here."""
    import torch.distributed as dist
    from . import nccl_event_cache_known_off
    if not capture_collectives:
        return "capture_collectives is off"
    if dist.get_backend(group) != "nccl":
        return "backend %s runs its collectives on the host" % dist.get_backend(group)
    if torch.device(device).type != "cuda":
        return "not a GPU device"
    # torch's NCCL event cache hands an event that a finished eager collective's Work still references to the next collective;
    # when that one is being CAPTURED the watchdog's query of the old Work throws hipErrorCapturedEvent and terminates the
    # process (reproduced: profiles/r04_notes.md).  The cache is read when a ProcessGroupNCCL is CONSTRUCTED: only a process whose
    # groups were all built with TORCH_NCCL_CUDA_EVENT_CACHE=0 may record (bbb_hip/__init__.py keeps the evidence).
    if not nccl_event_cache_known_off():
        return "a process group may predate TORCH_NCCL_CUDA_EVENT_CACHE=0 (import bbb_hip, or export it, before init_process_group)"
    if not _spans_world(group):
        return "the group is a strict sub-group of WORLD (private lane communicators need every rank in new_group)"
    return None


def collective_capture_ok(group, device, _force_probe=False, _stall_s=0.0):
    """Can this process group's all_gather_into_tensor be recorded into a hipGraph?  Decided ONCE per group and AGREED by all
    ranks: every rank -- whatever its own preconditions or its own probe said -- enters the same agreement all_reduce (MIN of
    "I can"), so no rank is left in a collective the others never enter.  Only "nccl" (= RCCL) is probed: a 4-float gather on a
    PRIVATE communicator (lane_group), captured on a side stream, replayed twice, result checked.  The probe's collectives run
    under a wall-clock watchdog (capture_probe_timeout_s): a stall yields the eager protocol, not a hang.
    (_force_probe / _stall_s: test hooks -- run the agreement protocol on a host backend too / sleep before entering it.)"""
    import torch.distributed as dist
    try:
        if group in _capture_probe:
            return _capture_probe[group]
        backend = dist.get_backend(group)
    except (TypeError, ValueError, RuntimeError) as exc:
        # not a process group torch knows (a stand-in object in a simulation, a destroyed group): nothing to record into
        last_protocol.update(collective="eager", reason="no usable process group: %s" % type(exc).__name__)
        return False
    reason = _local_capture_preconditions(group, device)
    gloo_like = backend != "nccl" and not _force_probe
    if gloo_like:
        # host-side backends never record; every rank knows that without talking (the backend is a property of the group)
        _capture_probe[group] = False
        last_protocol.update(collective="eager", reason=reason)
        return False
    if not _spans_world(group):
        # (sub-groups: lane_group cannot be entered -- every rank of the sub-group knows it: no probe, no agreement needed)
        _capture_probe[group] = False
        last_protocol.update(collective="eager", reason=reason)
        return False

    cancelled = []                                   # set when the watchdog gave up: a late helper thread must not start talking

    def agree(mine, pg):
        # ALWAYS entered by every rank, outside every try/except that swallows errors -- on a PRIVATE agreement communicator, so
        # that a rank that stalls here (the watchdog then abandons this thread inside the collective) leaves the caller's own
        # group untouched for the eager protocol
        if cancelled:
            raise RuntimeError("probe abandoned")
        flag = torch.tensor([1.0 if mine else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
        return bool(flag.item() > 0.5)

    def probe():
        if _stall_s:
            import time
            time.sleep(_stall_s)
        if cancelled:
            raise RuntimeError("probe abandoned")
        apg = lane_group(group, "agree")
        # phase 1: do ALL ranks meet their local preconditions?  (a rank that does not must not leave the others alone in new_group)
        if not agree(reason is None, apg):
            return False, reason or "another rank's preconditions do not allow recording"
        # phase 2: every rank records + replays a 4-float gather on a private communicator, then the ranks agree on the outcome
        ok, why = True, None
        try:
            pg = lane_group(group, "probe")                                   # created collectively by all ranks
            world, rank = dist.get_world_size(pg), dist.get_rank(pg)
            send = torch.full((4,), float(rank + 1), device=device)
            recv = torch.zeros((4 * world,), device=device)
            side = torch.cuda.Stream(device=device)
            g = torch.cuda.CUDAGraph()
            try:
                with ops.graph_capture(g, side):
                    dist.all_gather_into_tensor(recv, send, group=pg)
            except Exception as exc:                                          # a backend that refuses capture: fall back, do not fail
                ok, why = False, "capture refused: %s" % type(exc).__name__
            if ok:
                for v in (5.0, 9.0):
                    send.fill_(v + rank)
                    recv.zero_()
                    torch.cuda.synchronize(device)
                    with torch.cuda.stream(side):
                        g.replay()
                    side.synchronize()
                    want = torch.arange(world, device=device, dtype=torch.float32).repeat_interleave(4) + v
                    if not bool(torch.equal(recv, want)):
                        ok, why = False, "a replayed all_gather returned wrong data"
        except Exception as exc:                                              # noqa: BLE001
            ok, why = False, "probe failed: %s: %s" % (type(exc).__name__, str(exc)[:120])
        agreed = agree(ok, apg)
        if ok and not agreed:
            why = "another rank cannot record its collectives"
        return agreed, why

    done, res = _with_watchdog(probe, capture_probe_timeout_s)
    if done:
        ok, why = res
    else:
        cancelled.append(True)
        ok = False
        why = ("the capture probe did not finish within %.0f s (a rank is missing or stalled)" % capture_probe_timeout_s) if res is None \
            else "probe raised %s: %s" % (type(res).__name__, str(res)[:120])
    # Confirmation (advisor r05): the decision above is LOCAL where a watchdog fired -- a rank whose watchdog expires just as the last
    # agreement completes on its peers would cache "eager" while they cache "recorded", and the next step would hang on mismatched
    # protocols.  So every rank enters one more MIN agreement, on the caller's own group (the private agreement communicator may
    # hold an abandoned thread), again under the watchdog: "recorded" only if every rank says so and the round itself completes.
    def confirm():
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item() > 0.5)

    done2, res2 = _with_watchdog(confirm, capture_probe_timeout_s)
    if not done2:
        ok, why = False, why or "the confirmation round did not finish within %.0f s" % capture_probe_timeout_s
    elif ok and not res2:
        ok, why = False, "another rank fell back to the eager protocol (its probe timed out)"
    _capture_probe[group] = ok
    last_protocol.update(collective="recorded" if ok else "eager", reason=None if ok else why)
    return ok


class GraphedMC:
    """One Monte-Carlo step captured as a hipGraph (launch-bound inner loop -> one graph launch per step).

    The step is `mc_forward(net, x, num_ens, group=..., streams=...)` for a FIXED input buffer `x` (copy new batches
    into `self.x`), inference only.  Fresh noise on every replay: the kernels add a device-side counter to their call
    index and the graph itself increments that counter, so replay r uses exactly the calls r eager steps would have
    used; the Python-side counter is advanced in step().
    `lane` / `lanes`: this graph is lane `lane` of `lanes` graphs replayed round-robin (GraphedPipeline): its counter
    starts at lane*num_ens and advances by lanes*num_ens.
    With a process group the graph holds this rank's work units (shard_plan: (draw x batch-slice) units, or whole draws when
    the fast path does not apply) and ends by packing its [B, C] log-sum-exp block and its share of the KL sum into a
    preallocated send buffer.  When the backend's collectives can be recorded into a hipGraph (RCCL: collective_capture_ok,
    probed once per group and agreed on by all ranks) the ONE all_gather of the step and the reduction over ranks are part of
    the same graph: one host call per step.  Otherwise (gloo; a refused capture) step() issues the all_gather eagerly on the
    lane's stream and replays a second small graph for the reduction -- three host calls per step, which matters when 8 ranks
    leave each GPU only ~0.1 ms of work per step.
    step() returns (log_outputs [B, C], kl): buffers overwritten by the lane's next replay.
    steps > 1 (batch-innermost path): the graph holds `steps` consecutive steps -- `steps` batches, each with
    its own num_ens weight draws and its own noise calls (step g, draw j = call g * num_ens + j), exactly what `steps` separate
    replays would compute -- as ONE set of launches of steps * num_ens slabs (a one-draw step of a small model is a chain of ~10
    launch-latency-bound kernels; and the 10-draw launches of the metric step are one or two rounds of workgroups whose ramp and
    tail cost ~13 % of the GEMM and ~10 us of fixed cost in the parameter pass: `steps` of them per launch amortise both).  step(x) then only stores the batch in the next slot and replays when the
    last slot is filled (flush() replays a partly filled group); the (log_outputs, kl) it returns are that slot's views of the
    graph's output, valid once the group has been replayed and the lane's stream synchronised.
    steps > 1 WITH a process group: the steps * num_ens draws of the group are dealt to the ranks as contiguous ranges of whole
    draws on whole batches (group_share) -- single-device-sized launches on every rank instead of batch slices -- and ONE
    all_gather of [steps * B * C + 1] floats per group combines them; every rank gets every step's result; all ranks must call
    step() / flush() in the same sequence."""

    def __init__(self, net, x, num_ens, streams=1, kl_mode="sum", lane=0, lanes=1, stream=None, seed_call=None, group=None,
                 precision="fp32", steps=1, launch_config=None):
        _lib.require_device(x)
        self.steps, self.slot, self.lanes = int(steps), 0, int(lanes)
        # the launch-shape / arithmetic-mode choices this graph is planned and captured under: a snapshot, so that later changes of
        # the process defaults (or another pipeline built with other modes) never reach into this one
        self.launch_config = (launch_config if launch_config is not None else ops.current_config()).copy(launches_overlap=int(lanes) > 1)
        if self.steps > 1:
            with torch.no_grad():
                if not units_ok(net, x):
                    raise _lib.BBBHipError("steps > 1 needs the batch-innermost path (model / input shape not covered)")
            self.B = x.shape[0]
            x = x.repeat(self.steps, 1, 1, 1)                  # the group's input buffer: slot g = rows [g*B, (g+1)*B)
        # a lane of a pipeline gets its OWN input buffer: several steps are in flight, so the next batch must not be
        # written into memory an earlier step is still reading
        self.net, self.x, self.num_ens, self.group, self.kl_mode = net, (x.clone() if int(lanes) > 1 else x), int(num_ens), group, kl_mode
        self.precision = precision
        self.world = 1 if group is None else torch.distributed.get_world_size(group)
        rank = 0 if group is None else torch.distributed.get_rank(group)
        rng.assign_stream_ids(net)
        import os as _os
        self._force_combine = group is not None and _os.environ.get("BBB_FORCE_COMBINE") == "1"   # test hook: N > 1 code path at world 1
        self.multi = self.world > 1 or self._force_combine
        if self.steps > 1 and self.multi:
            # a group of steps over several ranks: contiguous ranges of the group's steps * num_ens draws (group_share)
            self.S = 1
            self.lo, self.hi, self.g_lo, self.n_gl, self.g_off = group_share(self.num_ens, self.steps, rank, self.world)
        else:
            with torch.no_grad(), ops.use_config(self.launch_config):   # the captured step is inference: plan for the inference path
                self.S, self.lo, self.hi = shard_plan(net, self.x, self.num_ens, rank, self.world, True, precision)
        dev = x.device
        self.stride = int(lanes) * self.num_ens * self.steps
        self.start = int(lane) * self.num_ens * self.steps
        self.counter = torch.full((1,), self.start, dtype=torch.int32, device=dev)
        self.seed, self.call0 = seed_call if seed_call is not None else rng.next_calls(0)
        self.own_stream = stream is not None
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.lse = self.kl_local = None
        self.shape = (output_rows(net, tuple(x.shape)) // self.steps, getattr(net, "num_classes", None))
        if self.steps > 1 and self.multi:
            self.shape = (self.shape[0] * self.steps, self.shape[1])       # the collective carries the whole group's rows
        self.fused = False                           # the collective and the reduction over ranks live inside the step's graph
        if self.multi:
            if self.shape[1] is None:
                raise _lib.BBBHipError("a sharded GraphedMC needs net.num_classes")
            n = self.shape[0] * self.shape[1] + 1
            self.world_size_for_buffers = self.world
            self.send = torch.full((n,), -float("inf"), dtype=torch.float32, device=dev)     # a rank without work sends -inf / 0
            self.send[-1] = 0.0
            self.recv = torch.empty((self.world * n,), dtype=torch.float32, device=dev)
        if self.multi:
            self.out_lo = torch.empty(self.shape, dtype=torch.float32, device=dev)
            self.out_kl = torch.empty((), dtype=torch.float32, device=dev)
            self.recv.zero_()
        can_fuse = self.multi and collective_capture_ok(group, dev)
        if can_fuse:
            self.group = lane_group(group, int(lane))            # a PRIVATE communicator per lane, lane 0 included: a recorded collective
                                                                 # never shares one with the caller's eager ones (every rank builds the
                                                                 # same lanes in the same order)
        self.protocol = "recorded" if can_fuse else ("eager" if self.multi else "none")
        if self.hi > self.lo or can_fuse:
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.no_grad(), torch.cuda.stream(self.stream), rng.device_call_offset(self.counter):
                for _ in range(2):                   # warm-up on the capture stream (allocator, lazy module state)
                    if self.hi > self.lo:
                        self._step_body(streams)
                    if can_fuse:
                        self._post_body()            # (no eager collective here: see _eager_gather; the communicator exists)
                self.counter.fill_(self.start)
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), rng.device_call_offset(self.counter), ops.graph_capture(self.graph, self.stream):
                if self.hi > self.lo:
                    self.lse, self.kl_local = self._step_body(streams)
                if can_fuse:
                    # ONE graph = this rank's units + the step's one collective + the reduction over ranks: one host call per step
                    _all_gather(self.recv, self.send, self.group)
                    self._post_body()
            self.fused = can_fuse
        else:
            self.graph = None                        # more ranks than draws: this rank only joins the collective
        self.replays = 0
        if self.multi and not self.fused:
            self.comm_stream = torch.cuda.Stream(device=dev)     # the eager all_gather of every step runs here (_eager_gather)
            # the reduction over ranks as its own small graph: gathered [world, B*C + 1] -> log_outputs, kl
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.no_grad(), torch.cuda.stream(self.stream):
                self._post_body()
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            torch.cuda.synchronize(dev)
            self.post = torch.cuda.CUDAGraph()
            with torch.no_grad(), ops.graph_capture(self.post, self.stream):
                self._post_body()

    def _post_body(self):
        n = self.send.numel()
        g = self.recv.view(self.world_size_for_buffers, n)
        blocks = g[:, :-1].reshape(self.world_size_for_buffers, *self.shape)
        self.out_lo.copy_(torch.logsumexp(blocks, dim=0) - math.log(self.num_ens))
        kl = g[:, -1].sum()
        if self.steps > 1:
            kl = kl / self.steps                     # the ranks' shares add up to the KL of steps * num_ens forwards
        self.out_kl.copy_(kl if self.kl_mode == "sum" else kl / self.num_ens)

    def _step_body(self, streams):
        # (launch_config incl. launches_overlap: choices that depend on what runs beside the launch; scratch owned by this graph)
        with ops.use_config(self.launch_config), ops.scratch_scope(self):
            return self._step_body_inner(streams)

    def _step_body_inner(self, streams):
        n_loc = self.hi - self.lo
        single = self.world == 1 and not self._force_combine
        if single:
            scale = float(self.num_ens) if self.kl_mode == "sum" else 1.0
        else:
            scale = float(n_loc) / self.S
        # the tail launch of the fast path also scales the KL and advances the noise counter (two element-wise launches less)
        end = [scale, self.counter, self.stride, False]
        if self.steps > 1 and self.multi:
            B = self.B
            lse, kl = _local_lse(self.net, self.x[self.g_lo * B:(self.g_lo + self.n_gl) * B], n_loc, self.seed, self.call0 + self.lo, 0,
                                 precision=self.precision, step_end=end, share=(self.num_ens, self.g_off))
        elif self.steps > 1:
            lse, kl = _local_lse(self.net, self.x, self.num_ens, self.seed, self.call0, self.num_ens, precision=self.precision,
                                 step_end=end, groups=self.steps)
        elif self.S > 1:
            lse, kl = _local_lse(self.net, self.x, self.num_ens, self.seed, self.call0, 0, precision=self.precision,
                                 units=(self.S, self.lo, self.hi), step_end=end)
        else:
            lse, kl = _local_lse(self.net, self.x, n_loc, self.seed, self.call0 + self.lo, self.num_ens if single else 0,
                                 streams=streams, precision=self.precision, step_end=end)
        if not end[3]:
            kl = kl * scale
            self.counter.add_(self.stride)           # part of the graph: next replay of this lane
        if self.multi:                               # pack what this rank contributes to the step's one collective
            if self.steps > 1:
                n = lse.shape[0] // self.n_gl * lse.shape[1]         # floats per step; steps without a local draw stay -inf
                self.send[self.g_lo * n:(self.g_lo + self.n_gl) * n].copy_(lse.reshape(-1))
            else:
                self.send[:-1].copy_(lse.reshape(-1))
            self.send[-1:].copy_(kl.reshape(1))
        return lse, kl

    def _launch(self):
        """One replay of the lane's graph (+ the eager collective protocol when the collective is not part of it)."""
        if self.graph is not None:
            self.graph.replay()
        self.replays += 1
        if self.multi and not self.fused:
            # ONE collective per launch (RCCL on GPUs), eager, on the lane's communication stream
            _eager_gather(self.recv, self.send, self.group, torch.cuda.current_stream(self.x.device), self.comm_stream)
            self.post.replay()

    def flush(self):
        """steps > 1: replay a partly filled group now (its empty slots recompute the batches they still hold)."""
        if self.steps > 1 and self.slot > 0:
            ctx = torch.cuda.stream(self.stream) if self.own_stream else _null_ctx()
            with ctx:
                self._launch()
            rng.next_calls((self.steps - self.slot) * self.num_ens)   # the graph consumed the calls of the empty slots too
            self.slot = 0

    def step(self, x=None):
        """Replay the step; `x` (optional) is copied into this lane's input buffer first, on the lane's stream."""
        producer = torch.cuda.current_stream(self.x.device) if (x is not None and self.own_stream) else None
        ctx = torch.cuda.stream(self.stream) if self.own_stream else _null_ctx()
        with ctx:
            if self.steps > 1:
                g, B = self.slot, self.B
                if x is not None and self.multi and not (self.hi > self.lo and self.g_lo <= g < self.g_lo + self.n_gl):
                    x = None                                       # a batch none of this rank's draws reads
                if x is not None:
                    if producer is not None:
                        self.stream.wait_stream(producer)
                    self.x[g * B:(g + 1) * B].copy_(x, non_blocking=True)
                self.slot += 1
                rng.next_calls(self.num_ens)
                if self.slot == self.steps:
                    self._launch()
                    self.slot = 0
                if self.multi:
                    return self.out_lo[g * B:(g + 1) * B], self.out_kl
                return self.lse[g * B:(g + 1) * B], self.kl_local
            if x is not None:
                if producer is not None:
                    self.stream.wait_stream(producer)              # whoever produced x on the caller's stream
                self.x.copy_(x, non_blocking=True)
            self._launch()
            rng.next_calls(self.num_ens)             # keep the host-side counter in step with the device's
            if not self.multi:
                return self.lse, self.kl_local
            return self.out_lo, self.out_kl


class GraphedLogits:
    """K stochastic forwards of `net` on one batch as a captured hipGraph -> logits [K, B', C] (contiguous, API layout): what
    the drop-in `net(x)` serves a speculated K-draw batch from (layers/_fused.py) -- ~30 launches through ctypes become one graph
    launch.  The captured kernels read a device-side call offset, so run(x, call) computes draws call .. call + K - 1 of the
    noise stream (seed fixed at capture), bit for bit what _mc_logits_chwn(net, x, K, seed, call) launches eagerly.  The output
    buffer belongs to the graph and is overwritten by the next run(): callers clone what they keep."""

    def __init__(self, net, x, K, precision="fp32", launch_config=None):
        _lib.require_device(x)
        self.K, self.precision = int(K), precision
        self.launch_config = (launch_config if launch_config is not None else ops.current_config()).copy()
        self.x = x.detach().clone()
        self.seed, self.call0 = rng.next_calls(0)
        dev = x.device
        self.counter = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.graph = None
        # captured on a throw-away stream, replayed on the caller's current stream: the scratch its launches record (KL partial
        # slots, split-contraction tickets) therefore belongs to THIS graph (ops.scratch_scope), not to the capture stream -- whose
        # handle torch's stream pool may hand to a later capture while this graph is still replayed elsewhere
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side), rng.device_call_offset(self.counter), ops.scratch_scope(self):
            for _ in range(2):                       # warm-up on a side stream (allocator, lazy module state, scratch growth)
                if self._body(net) is None:
                    return                           # this model / input does not fit the batch-innermost path
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), rng.device_call_offset(self.counter), ops.graph_capture(g, side), ops.scratch_scope(self):
            self.logits, self.kl = self._body(net)
        self.graph = g

    def _body(self, net):
        with ops.use_config(self.launch_config):
            out = _mc_logits_chwn(net, self.x, self.K, self.seed, self.call0, precision=self.precision)
        if out is None:
            return None
        return out[0].permute(0, 2, 1).contiguous(), out[1]

    def run(self, x, call):
        """Draws call .. call + K - 1 on batch x (enqueued on the current stream) -> (logits [K, B', C], kl): graph-owned buffers."""
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        d = (int(call) - self.call0) & 0xFFFFFFFF
        self.counter.fill_(d - (1 << 32) if d >= (1 << 31) else d)     # the kernels add it modulo 2^32
        self.graph.replay()
        return self.logits, self.kl


def _p2(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class GraphedPipeline:
    """`depth` independent Monte-Carlo steps in flight: GraphedMC lanes on their own HIP streams, replayed round-robin.
    Consecutive steps of an inference / validation loop do not depend on each other, so step i+1's kernels may run
    while step i's are still draining: each layer launch of an E=10 step is a single wave of workgroups, and the other
    step's kernels fill its ramp, its tail and the CUs its imbalance leaves idle.  Every step still does all of its
    work with its own noise (lane l, replay r = noise calls of step r*depth + l).
    step() returns the (log_outputs, kl) buffers of the lane just enqueued; call sync() (or synchronize the device)
    before reading them.
    steps_per_launch = G > 1: every lane's graph holds G consecutive steps (GraphedMC steps=G): step i goes to
    slot i % G of lane (i // G) % depth and the lane replays when its G slots are filled -- same noise calls per step as G = 1;
    sync() replays partly filled groups first."""

    def __init__(self, net, x, num_ens, depth=2, streams=1, kl_mode="sum", group=None, precision="fp32", steps_per_launch=1,
                 launch_config=None):
        seed_call = rng.next_calls(0)
        self.launch_config = (launch_config if launch_config is not None else ops.current_config()).copy()
        # lane streams come from a per-device pool and are REUSED by later pipelines: HIP maps streams onto a handful of hardware
        # queues round-robin, and a process that keeps creating streams (one pipeline per configuration, as bench.py does) ends up
        # with lanes that share a queue and stop overlapping (measured: the same 3-lane pipeline 15-18 % slower when built late)
        pool = _lane_streams(x.device, depth)
        self.G = int(steps_per_launch)
        self.lanes = [GraphedMC(net, x, num_ens, streams=streams, kl_mode=kl_mode, lane=l, lanes=depth,
                                stream=pool[l], seed_call=seed_call, group=group, precision=precision, steps=self.G,
                                launch_config=self.launch_config)
                      for l in range(depth)]
        self.i = 0
        self.dev = x.device

    def step(self, x=None):
        """Enqueue the next step on the next lane.  Pass the batch as `x` (copied into that lane's own buffer); without it
        the lane re-uses the batch it already holds."""
        lane = self.lanes[(self.i // self.G) % len(self.lanes)]
        self.i += 1
        return lane.step(x)

    def sync(self):
        if self.G > 1:
            # a partly filled group: only the lane currently being filled can have one; later steps start a fresh group
            for lane in self.lanes:
                lane.flush()
            self.i = -(-self.i // self.G) * self.G
        for lane in self.lanes:
            lane.stream.synchronize()
