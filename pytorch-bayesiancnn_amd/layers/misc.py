"""ModuleWrapper / FlattenLayer with the reference's semantics (layers/misc.py:4-35 upstream).

ModuleWrapper.forward: run the registered children in definition order, then return
(output, sum of kl_loss() over every sub-module that has one) -- the models define no forward of their
own, so attribute order is the graph.
"""
from torch import nn

_fast_inference = [True]


class reference_layout:
    """Context: whole-model forwards take the reference-layout (NCHW) kernels even where the batch-innermost inference path
    would apply (tests that compare the two paths; the results agree to the tolerance of the fused activation epilogue)."""

    def __enter__(self):
        self.prev = _fast_inference[0]
        _fast_inference[0] = False
        return self

    def __exit__(self, *exc):
        _fast_inference[0] = self.prev
        return False


def _refuse_low_precision(why):
    """ops.LaunchConfig.dropin_precision = "bf16" asks for the bf16 inference path: a forward that cannot take it must not quietly
    compute in fp32."""
    from bbb_hip import _lib, ops, rng
    if ops.current_config().dropin_precision != "fp32" and not rng._scope:
        raise _lib.BBBHipError("dropin_precision=%r: %s -- bf16 covers whole-model inference forwards of BBB models under "
                               "torch.no_grad() only" % (ops.current_config().dropin_precision, why))


class ModuleWrapper(nn.Module):
    """nn.Module with recursive flags and the universal (x) -> (x, kl) forward."""

    def set_flag(self, flag_name, value):
        setattr(self, flag_name, value)
        for child in self.children():
            setter = getattr(child, "set_flag", None)
            if setter is not None:
                setter(flag_name, value)

    def forward(self, x):
        from . import _fused
        if _fast_inference[0]:
            out = _fused.fast_forward(self, x)      # inference on the batch-innermost kernels (what mc_forward runs), E = 1
            if out is not None:
                return out
        _refuse_low_precision("this forward runs layer by layer (autograd enabled, hooks, replayed noise, a layout the batched path "
                              "does not cover, or reference_layout())")
        scope = None
        try:
            scope = _fused.enter(self)      # one noise call index (and, when possible, ONE fused reparam+KL
            y = _fused.hooked_chain(self, x, scope) if (_fast_inference[0] and scope is not None) else None
            if y is not None:               # children with forward hooks: the same kernels without the per-layer layout round trips
                x = y
            else:
                for child in self.children():   # launch) for every Bayesian layer below this wrapper
                    x = child(x)
            if scope is not None and scope.kl is not None:
                return x, scope.kl          # KL of all layers, already reduced on the device
            kl = 0.0
            for mod in self.modules():      # self included, like the reference
                if hasattr(mod, "kl_loss"):
                    kl = kl + mod.kl_loss()
            return x, kl
        finally:
            _fused.leave(scope)


class FlattenLayer(ModuleWrapper):
    """x.view(-1, num_features): [B,128,1,1] -> [B,128]; a 224x224 AlexNet input gives [B*49, 128]."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features

    def forward(self, x):
        return x.view(-1, self.num_features)
