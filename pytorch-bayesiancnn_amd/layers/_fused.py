"""Whole-model fusion behind ModuleWrapper.forward.

The reference runs, per layer and per forward, ~15 small ATen kernels for eps / softplus / mu+sigma*eps / KL.
When a model-level wrapper starts a forward we instead (a) reserve ONE noise call index for all its layers
and (b) launch the fused reparam+KL kernel ONCE over every layer's (W, bias) tensors, handing each layer its
sampled weights (BBB) or sigma^2 tensors (LRT) and returning the total KL as one device scalar.
Falls back to per-layer launches (same numbers) when a layer replays external eps or priors differ.
"""
import torch

from bbb_hip import rng


class _Scope:
    __slots__ = ("kl", "pushed", "layers")

    def __init__(self):
        self.kl = None
        self.pushed = False
        self.layers = ()


def enter(wrapper):
    from ._base import BayesianLayer
    if isinstance(wrapper, BayesianLayer):
        return None
    layers = [m for m in wrapper.modules() if isinstance(m, BayesianLayer)]
    if not layers:
        return None
    sc = _Scope()
    sc.layers = layers
    seed, call = rng.push_forward_scope()
    sc.pushed = True
    try:
        _presample(sc, layers, seed, call)
    except BaseException:
        leave(sc)                        # an aborted enter must not leave the noise scope pushed
        raise
    return sc


def _presample(sc, layers, seed, call):
    dev_ok = all(l.W_mu.is_cuda for l in layers)
    replay = any(l.eps_source is not None for l in layers)
    if dev_ok and not replay:
        from bbb_hip import ensemble
        from .bbb import _BBBLayer
        from .lrt import _LRTLayer
        bbb = [l for l in layers if isinstance(l, _BBBLayer)]
        lrt = [l for l in layers if isinstance(l, _LRTLayer)]
        kl = None
        if bbb:
            sampled, kl = ensemble._sample_all(bbb, 1, seed, call)
            for l, wb in sampled.items():
                l._presampled = wb
                l._kl = None
        if lrt:
            variances, k2 = ensemble._variances_all(lrt)
            for l, v in variances.items():
                l._presampled = v
                l._kl = None
            kl = k2 if kl is None else kl + k2
        sc.kl = kl


def leave(scope):
    if scope is not None and scope.pushed:
        scope.pushed = False
        rng.pop_forward_scope()
        for l in scope.layers:           # a forward that aborted midway must not hand stale samples to the next one
            l._presampled = None


def fast_forward(wrapper, x):
    """Whole-model inference forward on the batch-innermost path of bbb_hip.ensemble (pixel-major GEMMs that skip padding
    taps, activation fused into the epilogue, HIP pooling): one draw, the same kernels -- hence the same bits -- as draw j
    of a batched mc_forward under the same call index.  Applies when nothing needs autograd (torch.no_grad(), or frozen
    parameters), the input is a CUDA [B, C, H, W] batch with B % 4 == 0, the model is made of modules that path knows, it
    ends in a Bayesian linear layer, no layer replays external noise, and this is the outermost wrapper of the forward.
    Returns (output, kl) or None (-> the reference-layout path below, which also serves training)."""
    from ._base import BayesianLayer
    if isinstance(wrapper, BayesianLayer) or rng._scope or not torch.is_tensor(x) or not x.is_cuda or x.dim() != 4:
        return None
    from bbb_hip import ensemble
    mods = ensemble.flat_children(wrapper)
    if not mods or not isinstance(mods[-1], BayesianLayer) or not hasattr(mods[-1], "out_features"):
        return None
    layers = ensemble.bayesian_layers(wrapper)
    if any(l.eps_source is not None for l in layers) or not all(l.W_mu.is_cuda for l in layers):
        return None
    if torch.is_grad_enabled() and ensemble.any_requires_grad(wrapper):
        # training / the reference's validate loop (which does not disable autograd): the same kernels behind ONE autograd node
        from bbb_hip import fast_train
        if not ensemble.fast_autograd or not fast_train.train_path_ok(wrapper, x):
            return None
        seed, call = rng.next_calls(1)
        logits, kl = fast_train.mc_logits_autograd(wrapper, x, 1, seed, call)
        for l in layers:
            l._kl = None
        return logits[0].t(), kl
    if not ensemble._chwn_ok(wrapper, x):
        return None
    seed, call = rng.next_calls(1)
    out = ensemble._mc_logits_chwn(wrapper, x, 1, seed, call)
    if out is None:
        rng.rewind((seed, call))                 # nothing was launched: give the call index back
        return None
    logits, kl = out                             # [1, C, B']
    for l in layers:
        l._kl = None
    return logits[0].t().contiguous(), kl
