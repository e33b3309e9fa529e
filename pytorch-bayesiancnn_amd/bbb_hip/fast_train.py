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

from . import _lib, ops, rng


def _layers():
    from layers.bbb import _BBBLayer, BBBConv2d, BBBLinear
    from layers.lrt import _LRTLayer
    from layers.misc import FlattenLayer
    return _BBBLayer, BBBConv2d, BBBLinear, _LRTLayer, FlattenLayer


def train_path_ok(net, x):
    """The fast autograd path covers: a 4-d CUDA fp32 batch with B % 4 == 0; a flat model (ensemble.flat_children) of BBB conv /
    linear layers sharing one prior, each optionally followed by ReLU / Softplus(1, 20) and then MaxPool2d (no padding, floor),
    and a FlattenLayer that keeps one row per image; stride-1 convolutions everywhere except the first layer; the model ends in
    a Bayesian linear layer; no eps replay."""
    from . import ensemble
    _BBBLayer, BBBConv2d, BBBLinear, _LRTLayer, FlattenLayer = _layers()
    if not torch.is_tensor(x) or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32 or x.shape[0] % 4 != 0:
        return False
    mods = ensemble.flat_children(net)
    if not mods or not isinstance(mods[-1], BBBLinear):
        return False
    first = True
    pri = set()
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, _LRTLayer):
            return False
        if isinstance(m, _BBBLayer):
            if m.eps_source is not None or not m.W_mu.is_cuda or not m.use_bias:
                return False
            pri.add((m.prior_mu, m.prior_sigma))
            if isinstance(m, BBBConv2d):
                if not first and ops._pair(m.stride) != (1, 1):
                    return False
                (ph, pw), (dh, dw) = ops._pair(m.padding), ops._pair(m.dilation)
                if dh * (m.kernel_size[0] - 1) < ph or dw * (m.kernel_size[1] - 1) < pw:
                    return False
                if not first and m.in_channels % 4 != 0:
                    return False
            elif not first and m.in_features % 4 != 0:
                return False
            first = False
            if i + 1 < len(mods) and ensemble._act_name(mods[i + 1]) is not None:
                i += 1
        elif isinstance(m, nn.MaxPool2d):
            k, s = m.kernel_size, m.stride
            if not (isinstance(k, int) and isinstance(s, int) and m.padding == 0 and m.dilation == 1 and not m.ceil_mode):
                return False
        elif isinstance(m, FlattenLayer):
            pass
        else:
            return False                     # a stand-alone activation or anything else: not on this path
        i += 1
    if len(pri) != 1 or len(ensemble.bayesian_layers(net)) * 2 > _lib.MAX_SEGMENTS:
        return False
    return ensemble.output_rows(net, tuple(x.shape)) == x.shape[0]


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
        ctx.x_nchw = x.detach()
        return logits, kl

    @staticmethod
    def backward(ctx, g_logits, g_kl):
        cfg, tape = ctx.cfg, ctx.tape
        mus, rhos, ids, pm, ps, x_shape = ctx.meta
        E, seed, call0 = cfg["draws"], cfg["seed"], cfg["call0"]
        B = x_shape[0]
        gws = [None] * len(mus)
        g = g_logits.contiguous() if g_logits is not None else None
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
            gws[2 * li + 1] = g_pre.sum(dim=(2, 3, 4))                    # bias gradient [E, Cout]
            Cin = x_in.shape[1]
            if Cin % 4 == 0:
                gw = ops.conv2d_chwn_weight_grad(g_pre, x_in, tuple(w5.shape), stride, padding, dilation)
            elif rec["first"]:
                # 3-channel first layer: its input is shared by all draws, so the draws stack into the GEMM's row dimension
                gw = ops.conv2d_chwn_weight_grad_shared_input(g_pre, ctx.x_nchw, tuple(w5.shape), stride, padding, dilation)
            else:
                gy = g_pre.permute(0, 4, 1, 2, 3).contiguous()           # [E, B, Cout, Ho, Wo]
                xn = x_in.permute(0, 4, 1, 2, 3).contiguous()
                gw = ops.conv2d_weight_grad(gy, xn, tuple(w5.shape), stride, padding, dilation)
            gws[2 * li] = gw.reshape(ws_shape(rec))
            if not rec["first"]:
                g = ops.conv2d_chwn_input_grad(g_pre, w5, (x_in.shape[2], x_in.shape[3]), padding, dilation)
            else:
                g = None
        gmu, grho = ops.reparam_kl_backward(mus, rhos, gws, g_kl, pm, ps, ids, seed, call0, E)
        out = [None, None]
        for a, b in zip(gmu, grho):
            out += [a, b]
        return tuple(out)


def ws_shape(rec):
    m = rec["layer"]
    E = rec["w"].shape[0]
    return (E,) + tuple(m.W_mu.shape)


def mc_logits_autograd(net, x, draws, seed, call0):
    """Differentiable batched forward: -> (logits [E, C, B] batch-innermost, kl of one forward), gradients flow to every
    layer's W_mu, W_rho, bias_mu, bias_rho.  Call only when train_path_ok(net, x)."""
    from . import ensemble
    _lib.require_device(x)
    params = []
    for l in ensemble.bayesian_layers(net):
        params += [l.W_mu, l.W_rho, l.bias_mu, l.bias_rho]
    cfg = dict(net=net, draws=int(draws), seed=seed, call0=call0)
    return _MCForward.apply(cfg, x, *params)
