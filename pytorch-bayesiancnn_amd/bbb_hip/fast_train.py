"""Autograd for the batch-innermost ensemble path (training extension).

`mc_logits_autograd(net, x, draws, seed, call0)` is the differentiable counterpart of ensemble._mc_logits_chwn for models made of
BBB (weight-space) layers: ONE autograd node for the whole batched forward.  Forward = the inference kernels (fused reparam+KL,
pixel-major GEMMs that skip padding taps with the activation in the epilogue, HIP pooling) recording each layer's input and
activated output; backward walks the layers in reverse with
  * `bbb_pool_act_bwd_chwn`            pooling + activation backward from the activated output alone,
  * `ops.conv2d_chwn_input_grad`       dgrad = the forward kernel on flipped, channel-transposed weights,
  * `ops.conv2d_chwn_weight_grad`      wgrad = the forward kernel with batch <-> channel roles swapped (padding taps skipped),
  * `ops.conv2d_chwn_weight_grad_shared_input`  the first layer (3-channel images): draws stacked into the GEMM rows over an
                                       im2col of the shared input, K split over the launch's draws,
  * `ops.reparam_kl_backward`          gradients of (mu, rho) from the gradients of the sampled weights + d loss / d KL (eps
                                       regenerated from the counter),
so no activation, pooling or convolution of the step runs through torch autograd.  What main_bayesian.py:43-58 does per batch:
E forwards, KL, log_softmax / logmeanexp (the small tail stays in torch), ELBO, backward.
"""
import torch
import torch.nn as nn

from . import _lib, ops


def _layers():
    from layers.bbb import _BBBLayer, BBBConv2d, BBBLinear
    from layers.lrt import _LRTLayer
    from layers.misc import FlattenLayer
    return _BBBLayer, BBBConv2d, BBBLinear, _LRTLayer, FlattenLayer


def train_path_ok(net, x):
    """The fast autograd path covers: a 4-d CUDA fp32 batch with B % 4 == 0; a flat model (ensemble.flat_children) of Bayesian
    conv / linear layers of ONE kind (all BBB or all BBB_LRT) sharing one prior, each optionally followed by ReLU /
    Softplus(1, 20) and then MaxPool2d (no padding, floor), and a FlattenLayer that keeps one row per image; stride-1
    convolutions and channel counts that are multiples of 4 everywhere except the first layer; the model ends in a Bayesian
    linear layer; no eps replay.  Returns "bbb", "lrt" or None."""
    from . import ensemble
    if not torch.is_tensor(x) or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32 or x.shape[0] % 4 != 0:
        return None
    # what never changes for a given model and input shape is analysed once (ensemble._structure); what can change between
    # calls (eps replay switched on, a layer moved off the device) is re-checked every time
    cache = ensemble._structure(net)["train"]
    key = tuple(x.shape)
    if key not in cache:
        cache[key] = _train_path_static(net, x)
    kind = cache[key]
    if kind is None:
        return None
    for m in ensemble.bayesian_layers(net):
        if m.eps_source is not None or not m.W_mu.is_cuda:
            return None
    return kind


def _train_path_static(net, x):
    from . import ensemble
    from layers.lrt import BBBConv2d as LRTConv2d, BBBLinear as LRTLinear
    _BBBLayer, BBBConv2d, BBBLinear, _LRTLayer, FlattenLayer = _layers()
    mods = ensemble.flat_children(net)
    if not mods or not isinstance(mods[-1], (BBBLinear, LRTLinear)):
        return None
    first = True
    pri = set()
    kinds = set()
    i = 0
    after_layer = False                  # the previous module was a Bayesian layer (+ its fused activation): a pool may follow
    while i < len(mods):
        m = mods[i]
        poolable, after_layer = after_layer, False
        if isinstance(m, (_BBBLayer, _LRTLayer)):
            kinds.add("lrt" if isinstance(m, _LRTLayer) else "bbb")
            if not m.use_bias:
                return None
            pri.add((m.prior_mu, m.prior_sigma))
            if isinstance(m, (BBBConv2d, LRTConv2d)):
                if not first and ops._pair(m.stride) != (1, 1):
                    return None
                (ph, pw), (dh, dw) = ops._pair(m.padding), ops._pair(m.dilation)
                if dh * (m.kernel_size[0] - 1) < ph or dw * (m.kernel_size[1] - 1) < pw:
                    return None
            elif m.in_features % 4 != 0:
                return None              # (a first linear layer too: its weight gradient takes the 4-aligned channel path)
            first = False
            after_layer = True
            if i + 1 < len(mods) and ensemble._act_name(mods[i + 1]) is not None:
                i += 1
        elif isinstance(m, nn.MaxPool2d):
            k, s = m.kernel_size, m.stride
            if not (isinstance(k, int) and isinstance(s, int) and m.padding == 0 and m.dilation == 1 and not m.ceil_mode):
                return None
            if not poolable:
                return None              # a leading pool, or a pool after a pool / flatten: the node's backward pairs pools with layers
        elif isinstance(m, FlattenLayer):
            pass
        else:
            return None                      # a stand-alone activation or anything else: not on this path
        i += 1
    if len(pri) != 1 or len(kinds) != 1 or len(ensemble.bayesian_layers(net)) * 2 > _lib.MAX_SEGMENTS:
        return None
    if ensemble.output_rows(net, tuple(x.shape)) != x.shape[0]:
        return None
    return next(iter(kinds))


def ws_shape(rec):
    m = rec["layer"]
    E = rec["w"].shape[0]
    return (E,) + tuple(m.W_mu.shape)


class _MCForward(torch.autograd.Function):
    """(x, W_mu0, W_rho0, b_mu0, b_rho0, W_mu1, ...) -> (logits [E, C, B] batch-innermost, kl of one forward)."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        from . import ensemble
        _BBBLayer, BBBConv2d, BBBLinear, _LRTLayer, FlattenLayer = _layers()
        net, E, seed, call0 = cfg["net"], cfg["draws"], cfg["seed"], cfg["call0"]
        mods = ensemble.flat_children(net)
        layers = [m for m in mods if isinstance(m, _BBBLayer)]
        mus, rhos, ids = [], [], []
        for l in layers:
            m, r, i = l._param_lists()
            mus += [t.detach() for t in m]
            rhos += [t.detach() for t in r]
            ids += i
        pm, ps = layers[0].prior_mu, layers[0].prior_sigma
        ws, _, kl = ops.reparam_kl_forward(mus, rhos, pm, ps, ids, seed, call0, draws=E)
        B = x.shape[0]
        h = ops.to_batch_innermost(x.detach()).unsqueeze(0)               # [1, C, H, W, B]
        tape = []                                                          # per Bayesian layer, in forward order
        li, i = 0, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _BBBLayer):
                w, b = ws[2 * li], ws[2 * li + 1]
                is_conv = isinstance(m, BBBConv2d)
                geom = (m.stride, m.padding, m.dilation) if is_conv else (1, 0, 1)
                x_in = h if is_conv else h.reshape(h.shape[0], m.in_features, 1, 1, B)
                w5 = w if is_conv else w.reshape(E, m.out_features, m.in_features, 1, 1)
                act = ensemble._act_name(mods[i + 1]) if i + 1 < len(mods) else None
                y = ops.conv2d_chwn_forward(x_in, w5, b, *geom, act=act)
                rec = dict(layer=m, x=x_in, w=w5, y=y, act=act, geom=geom, pool=None, first=(li == 0))
                if act is not None:
                    i += 1
                h = y
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.MaxPool2d):
                    pool = mods[i + 1]
                    h = ops.maxpool_chwn(y, pool.kernel_size, pool.stride)
                    rec["pool"] = (pool.kernel_size, pool.stride)
                    i += 1
                rec["out_shape"] = tuple(h.shape)
                tape.append(rec)
                li += 1
            elif isinstance(m, nn.MaxPool2d):                              # a pool that does not follow a Bayesian layer: on the input
                raise _lib.BBBHipError("fast_train: pooling must follow a Bayesian layer")
            elif isinstance(m, FlattenLayer):
                h = h.reshape(h.shape[0], m.num_features, 1, 1, B)
            i += 1
        logits = h.reshape(E, -1, B)
        ctx.cfg, ctx.tape, ctx.meta = cfg, tape, (mus, rhos, ids, pm, ps, tuple(x.shape))
        ctx.launch_config = ops.current_config()              # backward runs on autograd's device thread: carry the modes along
        ctx.scratch_token = ops.current_scratch_token()
        ctx.x_nchw = x.detach()
        ctx.versions = [(p, p._version) for p in params]      # backward re-reads the live (mu, rho): they must not have moved
        return logits, kl

    @staticmethod
    def backward(ctx, g_logits, g_kl):
        with ops.use_config(ctx.launch_config), ops.scratch_scope(token=ctx.scratch_token):
            return _MCForward._backward(ctx, g_logits, g_kl)

    @staticmethod
    def _backward(ctx, g_logits, g_kl):
        cfg, tape = ctx.cfg, ctx.tape
        _check_versions(ctx.versions)
        cfg["spent"] = True                                    # layers/_fused.py: no further draws are served from this graph
        mus, rhos, ids, pm, ps, x_shape = ctx.meta
        E, seed, call0 = cfg["draws"], cfg["seed"], cfg["call0"]
        B = x_shape[0]
        gws = [None] * len(mus)
        g = g_logits.contiguous() if g_logits is not None else None
        # a layer's weight / bias gradients and its input gradient are independent: the former run on a side stream beside the latter
        # (both are launches that leave the chip partly idle on their own); joined before the parameter pass's backward.  Tensors
        # the side stream reads stay referenced in `keep` until the join (the allocator must not hand them out again before).
        # (eager launches only: inside a captured step the forks cost more than the overlap gains -- 4.8 against 3.05 ms per step)
        side = _side_stream(g.device) if (g is not None and overlap_wgrad[0] and not torch.cuda.is_current_stream_capturing()) else None
        main = torch.cuda.current_stream(g.device) if side is not None else None
        keep = []
        # the first layer's im2col depends on the batch alone: on a side stream NOW, beside the small last layers' launches, instead of
        # in the tail of the chain where nothing else is left to run beside it
        xk_first = [None]
        r0 = tape[0]
        if side is not None and g is not None and r0["x"].shape[1] % 4 != 0:
            side.wait_stream(main)
            xk_stream = side.current
            with torch.cuda.stream(xk_stream):
                xk_first[0] = ops.im2col_pbj(ctx.x_nchw, tuple(r0["w"].shape), *r0["geom"])
            keep.append(xk_first[0])
        # every layer's input-gradient weights (flipped, channel-transposed: they depend on the sampled weights alone) in ONE launch up
        # front instead of one launch per layer on the chain -- for launch-bound steps (no side streams: captured, small).  A
        # many-draw step keeps them on the chain: up front, even on the side stream, they cost it 40 us (2.45 -> 2.49 ms at 512 x 10,
        # profiles/experiments/ab_flips_up_front.py)
        w_flipped = {}
        if flips_up_front[0] and side is None and g is not None and len(tape) > 1:
            outs = ops.flip_transpose_w_multi([tape[li]["w"] for li in range(1, len(tape))])
            w_flipped = {li: o for li, o in zip(range(1, len(tape)), outs)}
        for li in range(len(tape) - 1, -1, -1):
            rec = tape[li]
            y, w5, x_in, act = rec["y"], rec["w"], rec["x"], rec["act"]
            stride, padding, dilation = rec["geom"]
            if g is None:
                break
            g = g.reshape(rec["out_shape"])
            pad = rec["first"] and x_in.shape[1] % 4 != 0                # feeds conv2d_chwn_weight_grad_shared_input
            if rec["pool"] is not None:
                g_pre = ops.pool_act_backward_chwn(g, y, rec["pool"][0], rec["pool"][1], act, pad_planes=pad)
            elif act is not None:
                g_pre = ops.pool_act_backward_chwn(g, y, 0, 1, act, pad_planes=pad)
            else:
                g_pre = g
            def weight_side(g_pre=g_pre, x_in=x_in, w5=w5, rec=rec, li=li, stride=stride, padding=padding, dilation=dilation):
                gws[2 * li + 1] = ops.plane_sums(g_pre)                   # bias gradient [E, Cout]
                Cin = x_in.shape[1]
                if Cin % 4 == 0 or not rec["first"]:                     # (6-channel inputs etc.: padded to 8 inside)
                    gw = ops.conv2d_chwn_weight_grad(g_pre, x_in, tuple(w5.shape), stride, padding, dilation)
                else:
                    # 3-channel first layer: its input is shared by all draws, so the draws stack into the GEMM's row dimension
                    gw = ops.conv2d_chwn_weight_grad_shared_input(g_pre, ctx.x_nchw, tuple(w5.shape), stride, padding, dilation, xk=xk_first[0])
                gws[2 * li] = gw.reshape(ws_shape(rec))

            if side is not None and not rec["first"]:
                keep.append(g_pre)
                side.wait_stream(main)
                with torch.cuda.stream(side.current):
                    weight_side()
            else:
                if rec["first"] and xk_first[0] is not None:
                    main.wait_stream(xk_stream)
                weight_side()
            if not rec["first"]:
                g = ops.conv2d_chwn_input_grad(g_pre, w5, (x_in.shape[2], x_in.shape[3]), padding, dilation, w_flipped=w_flipped.get(li))
            else:
                g = None
        if side is not None:
            for st_ in side.streams:
                main.wait_stream(st_)
        del keep
        gmu, grho = ops.reparam_kl_backward(mus, rhos, gws, g_kl, pm, ps, ids, seed, call0, E)
        out = [None, None]
        for a, b in zip(gmu, grho):
            out += [a, b]
        return tuple(out)


fold_lrt_combine = [True]       # an LRT layer's g1 + 2 x g2 is formed inside the pooling / activation backward of the layer below
flips_up_front = [True]         # every layer's flipped input-gradient weights in one launch at the start of the backward
pair_lrt_backward = [True]      # an LRT layer's (mean, variance) gradient pairs as the two draws of one launch (see _MCForwardLRT._backward)
overlap_wgrad = [True]
_side_streams = {}


n_side_streams = [int(__import__("os").environ.get("BBB_TRAIN_SIDE_STREAMS", "2"))]      # (measured 1 / 2 / 3: 2.895 / 2.75 / 2.78 ms per bs 512 x 10 step)


class _Sides:
    """The side streams of a device, handed out round robin (layer by layer); waits / joins address all of them."""

    def __init__(self, device, n):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        self.i = 0

    def wait_stream(self, main):
        self.i = (self.i + 1) % len(self.streams)
        self.streams[self.i].wait_stream(main)

    @property
    def current(self):
        return self.streams[self.i]


def _side_stream(device):
    key = (torch.device(device).index, n_side_streams[0])
    if key not in _side_streams:
        _side_streams[key] = _Sides(device, n_side_streams[0])
    return _side_streams[key]


def _check_versions(versions):
    """What autograd's saved-tensor check would raise: a parameter (or sigma^2 operand) changed in place between the forward
    that recorded this node and its backward, which re-reads the live storage."""
    for p, v in versions:
        if p._version != v:
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                               "a Bayesian layer's parameter changed between the forward and this backward (optimizer.step() "
                               "before a delayed backward?)")


def _inverse_act(y, act):
    """Pre-activation from the activated output (what the LRT backward needs to recover sqrt(act_var) * eps = v - act_mu)."""
    if act == "softplus":
        return torch.where(y > 20.0, y, y + torch.log(-torch.expm1(-y)))
    return y                                  # ReLU: exact where y > 0; elsewhere the incoming gradient is zero anyway


class _MCForwardLRT(torch.autograd.Function):
    """Local-reparameterisation layers (layers/BBB_LRT/BBBConv.py:62-81): (x, W_mu0, W_var0, b_mu0, b_var0, ...) -> logits
    [E, C, B] batch-innermost.  Forward = the dual-accumulator LRT GEMM (act_mu, act_var, sample + activation in the epilogue);
    for the first layer, whose input and weights are the same for every draw, the two contractions run once and
    bbb_lrt_sample_chwn draws the E outputs.  Backward: out = act_mu + sqrt(act_var) * eps gives d/d act_mu = g and
    d/d act_var = g * (v - act_mu) / (2 act_var) with v recovered from the stored activated output; then two weight gradients
    (g_mu with x, g_var with x^2) and, below the first layer, g_x = dgrad(g_mu, W_mu) + 2 x * dgrad(g_var, W_var), all on the
    forward GEMM kernel (fast_train module docstring)."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        from . import ensemble
        from layers.lrt import BBBConv2d as LRTConv2d
        _BBBLayer, BBBConv2d, BBBLinear, _LRTLayer, FlattenLayer = _layers()
        net, E, seed, call0 = cfg["net"], cfg["draws"], cfg["seed"], cfg["call0"]
        mods = ensemble.flat_children(net)
        B = x.shape[0]
        h = ops.to_batch_innermost(x.detach()).unsqueeze(0)
        tape, li, i = [], 0, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _LRTLayer):
                w_mu, w_var, b_mu, b_var = (t.detach() for t in params[4 * li:4 * li + 4])
                is_conv = isinstance(m, LRTConv2d)
                geom = (m.stride, m.padding, m.dilation) if is_conv else (1, 0, 1)
                x_in = h if is_conv else h.reshape(h.shape[0], m.in_features, 1, 1, B)
                if not is_conv:
                    shp = (m.out_features, m.in_features, 1, 1)
                    w_mu, w_var = w_mu.reshape(shp), w_var.reshape(shp)
                act = ensemble._act_name(mods[i + 1]) if i + 1 < len(mods) else None
                sid = m._stream_base + 2
                shared = x_in.shape[0] == 1 and E > 1
                if shared:
                    _, am, av = ops.lrt_conv2d_chwn_forward(x_in, w_mu, w_var, b_mu, b_var, seed, call0, sid, *geom, sample=False,
                                                            want_moments=True, act=None)
                    y = ops.lrt_sample_chwn(am, av, E, seed, call0, sid, act=act)
                else:
                    y, am, av = ops.lrt_conv2d_chwn_forward(x_in, w_mu, w_var, b_mu, b_var, seed, call0, sid, *geom, sample=True,
                                                            want_moments=True, act=act)
                rec = dict(layer=m, x=x_in, w_mu=w_mu, w_var=w_var, y=y, am=am, av=av, act=act, geom=geom, pool=None, first=(li == 0))
                if act is not None:
                    i += 1
                h = y
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.MaxPool2d):
                    pool = mods[i + 1]
                    h = ops.maxpool_chwn(y, pool.kernel_size, pool.stride)
                    rec["pool"] = (pool.kernel_size, pool.stride)
                    i += 1
                rec["out_shape"] = tuple(h.shape)
                tape.append(rec)
                li += 1
            elif isinstance(m, nn.MaxPool2d):
                raise _lib.BBBHipError("fast_train: pooling must follow a Bayesian layer")
            elif isinstance(m, FlattenLayer):
                h = h.reshape(h.shape[0], m.num_features, 1, 1, B)
            i += 1
        ctx.cfg, ctx.tape = cfg, tape
        ctx.launch_config = ops.current_config()
        ctx.scratch_token = ops.current_scratch_token()
        ctx.x_nchw = x.detach()
        ctx.versions = [(p, p._version) for p in params]
        return h.reshape(h.shape[0], -1, B)

    @staticmethod
    def backward(ctx, g_logits):
        with ops.use_config(ctx.launch_config), ops.scratch_scope(token=ctx.scratch_token):
            return _MCForwardLRT._backward(ctx, g_logits)

    @staticmethod
    def _backward(ctx, g_logits):
        tape = ctx.tape
        _check_versions(ctx.versions)
        ctx.cfg["spent"] = True
        grads = [None] * (4 * len(tape))
        g = g_logits.contiguous()
        # (as in _MCForward: the weight-side work of a layer on a second stream beside its input gradients, eager launches only)
        side = _side_stream(g.device) if (overlap_wgrad[0] and not torch.cuda.is_current_stream_capturing()) else None
        main = torch.cuda.current_stream(g.device) if side is not None else None
        keep = []
        # (as in _MCForward: the flipped input-gradient weights of every layer -- mean and variance sets -- in one launch up front)
        w_flipped = {}
        if flips_up_front[0] and len(tape) > 1:
            pairs = [(tape[li]["w_mu"].unsqueeze(0), tape[li]["w_var"].unsqueeze(0)) for li in range(1, len(tape))]
            outs = ops.flip_transpose_w_multi(pairs)                     # [2, Cin, Cout, kh, kw] each: the mean set, then the variance set
            w_flipped = {li: o for li, o in zip(range(1, len(tape)), outs)}
        for li in range(len(tape) - 1, -1, -1):
            rec = tape[li]
            y, am, av, x_in, act = rec["y"], rec["am"], rec["av"], rec["x"], rec["act"]
            w_mu, w_var = rec["w_mu"], rec["w_var"]
            stride, padding, dilation = rec["geom"]
            # the incoming gradient: a tensor (the logits' gradient), or the layer above's two input gradients + this layer's output,
            # combined (g1 + 2 x g2) inside the pass below instead of by a launch of its own
            comb = None
            if isinstance(g, tuple):
                g, comb = g[0].reshape(rec["out_shape"]), (g[1].reshape((-1,) + tuple(rec["out_shape"][1:])), g[2].reshape(rec["out_shape"]))
            else:
                g = g.reshape(rec["out_shape"])
            # one pass: pooling / activation backward AND the split into d/d act_mu, d/d act_var (was ~10 ATen kernels per layer)
            k, s = rec["pool"] if rec["pool"] is not None else (0, 1)
            pad = rec["first"] and x_in.shape[1] % 4 != 0                # feeds conv2d_chwn_weight_grad_shared_input
            # below the first layer the pair (d/d act_mu, d/d act_var) stays ONE buffer: its two weight gradients (with x, x^2) and,
            # for a single draw, its two input gradients (with W_mu, W_var) run as the draws of one launch each
            g_pair = None
            if pair_lrt_backward[0] and not rec["first"]:
                g_pair = ops.lrt_pool_act_backward_chwn(g, y, am, av, k, s, act, stacked=True, combine=comb)        # [2, E, Cout, Ho, Wo, B]
                g_mu, g_var = g_pair[0], g_pair[1]
            else:
                g_mu, g_var = ops.lrt_pool_act_backward_chwn(g, y, am, av, k, s, act, pad_planes=pad, combine=comb)
            if am.shape[0] == 1 and g_mu.shape[0] > 1:      # first layer: one pair of moments feeds every draw
                g_mu, g_var = ops.sum_over_draws(g_mu, keepdim=True), ops.sum_over_draws(g_var, keepdim=True)
            def weight_side(g_mu=g_mu, g_var=g_var, x_in=x_in, w_mu=w_mu, rec=rec, li=li, stride=stride, padding=padding, dilation=dilation,
                            g_pair=g_pair):
                wshape = (1,) + tuple(w_mu.shape)
                Ed = g_mu.shape[0]
                if g_pair is not None and Ed == 1:
                    gb = ops.plane_sums(g_pair.reshape((2,) + tuple(g_mu.shape[1:])))               # [2, Cout]
                    grads[4 * li + 2], grads[4 * li + 3] = gb[0], gb[1]
                else:
                    grads[4 * li + 2] = ops.plane_sums(g_mu, over_draws=True)
                    grads[4 * li + 3] = ops.plane_sums(g_var, over_draws=True)
                if g_pair is not None and x_in.shape[0] == Ed:
                    gw = ops.conv2d_chwn_weight_grad(g_pair.reshape((2 * Ed,) + tuple(g_mu.shape[1:])), x_in,
                                                     (2 * Ed,) + tuple(w_mu.shape), stride, padding, dilation, x_squares=True)
                    gw_mu, gw_var = ops.sum_over_draws(gw[:Ed]), ops.sum_over_draws(gw[Ed:])
                elif x_in.shape[1] % 4 == 0 or not rec["first"]:
                    gw_mu = ops.conv2d_chwn_weight_grad(g_mu, x_in, wshape, stride, padding, dilation)
                    gw_var = ops.conv2d_chwn_weight_grad(g_var, ops.square(x_in), wshape, stride, padding, dilation)
                    gw_mu, gw_var = ops.sum_over_draws(gw_mu), ops.sum_over_draws(gw_var)
                else:
                    xk = ops.im2col_pbj(ctx.x_nchw, wshape, stride, padding, dilation)      # (its square = the im2col of x^2)
                    gw_mu = ops.conv2d_chwn_weight_grad_shared_input(g_mu, None, wshape, stride, padding, dilation, xk=xk)[0]
                    gw_var = ops.conv2d_chwn_weight_grad_shared_input(g_var, None, wshape, stride, padding, dilation, xk=ops.square(xk))[0]
                m = rec["layer"]
                grads[4 * li] = gw_mu.reshape(m.W_mu.shape)
                grads[4 * li + 1] = gw_var.reshape(m.W_mu.shape)

            if side is not None and not rec["first"]:
                keep += [g_mu, g_var]
                side.wait_stream(main)
                with torch.cuda.stream(side.current):
                    weight_side()
            else:
                weight_side()
            if not rec["first"]:
                hw = (x_in.shape[2], x_in.shape[3])
                w_t = w_flipped.get(li)
                if g_pair is not None and g_mu.shape[0] == 1:
                    if w_t is None:
                        w_t = ops.flip_transpose_w_pair(w_mu.unsqueeze(0), w_var.unsqueeze(0))
                    gx = ops.conv2d_chwn_input_grad(g_pair.reshape((2,) + tuple(g_mu.shape[1:])), w_mu.unsqueeze(0), hw, padding, dilation,
                                                    w_flipped=w_t)
                    g1, g2 = gx[0:1], gx[1:2]
                else:
                    t_mu, t_var = (w_t[0:1], w_t[1:2]) if w_t is not None else (None, None)
                    g1 = ops.conv2d_chwn_input_grad(g_mu, w_mu.unsqueeze(0), hw, padding, dilation, w_flipped=t_mu)
                    g2 = ops.conv2d_chwn_input_grad(g_var, w_var.unsqueeze(0), hw, padding, dilation, w_flipped=t_var)
                # (g1 + 2 x g2 is formed by the layer below's pooling / activation pass)
                g = (g1, x_in, g2) if fold_lrt_combine[0] else ops.lrt_input_grad_combine(g1, x_in, g2)
        if side is not None:
            for st_ in side.streams:
                main.wait_stream(st_)
        del keep
        return (None, None, *grads)


def mc_logits_autograd(net, x, draws, seed, call0, alias=None):
    """Differentiable batched forward: -> (logits [E, C, B] batch-innermost, kl of one forward), gradients flow to every
    layer's W_mu, W_rho, bias_mu, bias_rho.  Call only when train_path_ok(net, x).
    alias: {id(parameter): leaf tensor sharing its storage} -- the autograd graph is then rooted at those leaves instead of the
    parameters (train.GraphedTrainStep: fresh leaves have no gradient accumulator bound to another stream)."""
    def A(t):
        return t if alias is None else alias.get(id(t), t)

    from . import ensemble
    _lib.require_device(x)
    kind = train_path_ok(net, x)
    if kind == "lrt":
        layers = ensemble.bayesian_layers(net)
        mus, rhos = [], []
        for l in layers:
            m, r, _ = l._param_lists()
            mus += [A(t) for t in m]
            rhos += [A(t) for t in r]
        kl, s2, mu = ops.kl_only(mus, rhos, layers[0].prior_mu, layers[0].prior_sigma, want_sigma=True, sigma_squared=True,
                                 mu_through=True)
        flat = []
        for li, l in enumerate(layers):
            flat += [mu[2 * li], s2[2 * li], mu[2 * li + 1], s2[2 * li + 1]]
        cfg = dict(net=net, draws=int(draws), seed=seed, call0=call0)
        logits = _MCForwardLRT.apply(cfg, x, *flat)
        logits.bbb_cfg = cfg
        return logits, kl
    params = []
    for l in ensemble.bayesian_layers(net):
        params += [A(l.W_mu), A(l.W_rho), A(l.bias_mu), A(l.bias_rho)]
    cfg = dict(net=net, draws=int(draws), seed=seed, call0=call0)
    logits, kl = _MCForward.apply(cfg, x, *params)
    logits.bbb_cfg = cfg
    return logits, kl
