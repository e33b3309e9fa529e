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
    from bbb_hip import ensemble
    layers = ensemble.bayesian_layers(wrapper)           # (cached on the module: walking the tree costs ~10 us per forward)
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
                l.__dict__["_presampled"] = wb           # (plain attributes: nn.Module.__setattr__ costs ~2.5 us each)
                l.__dict__["_kl"] = None
        if lrt:
            variances, k2 = ensemble._variances_all(lrt)
            for l, v in variances.items():
                l.__dict__["_presampled"] = v
                l.__dict__["_kl"] = None
            kl = k2 if kl is None else kl + k2
        sc.kl = kl


def leave(scope):
    if scope is not None and scope.pushed:
        scope.pushed = False
        rng.pop_forward_scope()
        for l in scope.layers:           # a forward that aborted midway must not hand stale samples to the next one
            l.__dict__["_presampled"] = None


speculation = {"enabled": True, "max_draws": 32,
               # a speculated K-draw batch under no_grad is served from a captured hipGraph (ensemble.GraphedLogits) once the same
               # (input shape, K, noise seed, parameter storage, kernel modes) was seen `graph_after` times: one graph launch instead of
               # ~30 launches through ctypes.  At most `graphs` graphs per model (least recently used goes first).
               "graph_after": 2, "graphs": 4}


def _logit_graph(wrapper, st, x, K, seed, pver, precision="fp32"):
    """The cached GraphedLogits for this (model, input shape, K, seed, parameter storage, kernel modes), building it on the
    `graph_after`-th sighting; None = run eagerly (not seen often enough yet, capture not possible, or the path does not apply)."""
    from bbb_hip import ensemble, ops
    if not speculation.get("graph_after"):
        return None
    cache = st.setdefault("logit_graphs", {})
    key = (tuple(x.shape), x.dtype, x.device.index, int(K), int(seed), tuple(p[1] for p in pver[1:]), ops.current_config().key())
    ent = cache.get(key)
    if ent is None:
        ent = cache[key] = [0, None]
        while len(cache) > int(speculation["graphs"]):
            cache.pop(next(iter(cache)))                       # dicts keep insertion order: the oldest entry goes
    else:
        cache[key] = cache.pop(key)                            # most recently used last
    ent[0] += 1
    if ent[1] is None and ent[0] >= int(speculation["graph_after"]):
        # an optimisation of a forward the eager path serves anyway: whatever goes wrong while warming up / capturing (allocator,
        # a capture the runtime refuses) must not surface from the user's net(x) -- this key simply stays eager
        try:
            g = ensemble.GraphedLogits(wrapper, x, K, precision=precision)
            ent[1] = g if g.graph is not None else False
        except Exception as exc:                       # noqa: BLE001
            ent[1] = False
            st["logit_graph_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
    return ent[1] or None


class _Spec:
    """Speculative draw batching behind `net(x)` (main_bayesian.py:73-80 calls net(inputs) num_ens times on the SAME tensor):
    when a call repeats the previous call's input, the next K draws are computed in ONE batched launch -- call indices
    call .. call+K-1, i.e. exactly the noise the next K calls of the loop would use -- and those calls are served from the
    result (the generator offset still advances by one call per net(x)).  Bit-identical to the loop by the noise contract
    (draw j of a batched launch == net(x) number j).  K follows the length of the previous streak of identical inputs
    (a validation loop settles on K = num_ens at the FIRST call of each batch), or doubles 2, 4, 8 while no history exists.
    The cache is dropped when the input object, its version counter, any parameter's version counter, the generator
    position, the autograd mode or the model's structure (a replaced layer / Parameter, moved storage) changes."""
    __slots__ = ("xref", "xver", "pver", "grad", "seed", "next_call", "logits", "kl", "idx", "streak", "last_streak", "batches")

    def __init__(self):
        self.xref = None
        self.logits = None
        self.streak = 0
        self.last_streak = 0
        self.batches = 0

    # the cache lives on the wrapper (wrapper.__dict__): copies / pickles of the model (copy.deepcopy, torch.save(net)) start with
    # an empty one instead of dragging cached draws and a weak reference along
    def __deepcopy__(self, memo):
        return _Spec()

    def __reduce__(self):
        return (_Spec, ())


def _hooks_present(wrapper, mods):
    import torch.nn.modules.module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    if wrapper._forward_hooks or wrapper._forward_pre_hooks:
        return True
    for m in mods:
        if m._forward_hooks or m._forward_pre_hooks:
            return True
    return False


hooked_chain_enabled = [True]


def hooked_chain(wrapper, x, scope):
    """The per-layer forward of a model whose children carry forward (pre-)hooks, WITHOUT leaving the batch-innermost layout between
    the layers: the children run in definition order on the kernels the per-layer path uses anyway (ops.conv2d_layer /
    lrt_conv2d_layer: batch-innermost GEMM, plain fmaf chain; torch activations; HIP pooling) -- bit for bit that path's results --
    but the layout transposes around every layer are gone: only a module that carries a hook gets its input and output as the
    NCHW tensors the hook expects (and may replace).  ~19 launches per forward instead of ~35 (the path is host-bound).
    Applies to: a flat wrapper of Bayesian conv / linear layers, ReLU / Softplus(1, 20), MaxPool2d (no padding, floor) and a
    FlattenLayer that keeps one row per image; a 4-d fp32 GPU batch with B % 4 == 0; no autograd; plain hooks (no kwargs hooks, no
    global hooks); every layer's weights presampled by enter().  Returns the output, or None (-> the module-by-module loop)."""
    import torch.nn as nn
    import torch.nn.functional as F
    import torch.nn.modules.module as _m
    from bbb_hip import ensemble, ops
    from ._base import BayesianLayer
    from .bbb import _BBBLayer
    from .lrt import _LRTLayer
    from .misc import FlattenLayer
    if not hooked_chain_enabled[0] or scope is None or scope.kl is None:
        return None
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[0] % 4 == 0 and x.shape[0] > 0):
        return None
    if torch.is_grad_enabled() and (x.requires_grad or ensemble.any_requires_grad(wrapper)):
        return None
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return None
    mods = list(wrapper.children())
    if not mods or mods != ensemble.flat_children(wrapper) or not ensemble._chwn_ok(wrapper, x):
        return None
    if ensemble.output_rows(wrapper, tuple(x.shape)) != x.shape[0]:
        return None                                          # (the 224 x 224 flatten quirk: rows multiply -- the module-by-module loop)
    for m in mods:
        if getattr(m, "_forward_hooks_with_kwargs", None) or getattr(m, "_forward_pre_hooks_with_kwargs", None):
            return None
        if isinstance(m, BayesianLayer) and (m.eps_source is not None or m._presampled is None or not m.W_mu.is_cuda):
            return None
    B = x.shape[0]
    st = {"nchw": x, "chwn": None}                          # the running activation in either layout (at least one is set)

    def chwn():
        if st["chwn"] is None:
            t = st["nchw"]
            st["chwn"] = (ops.to_batch_innermost(t) if t.dim() == 4 else t.t().reshape(t.shape[1], 1, 1, B)).unsqueeze(0)
        return st["chwn"]                                    # [1, C, H, W, B]

    def nchw():
        if st["nchw"] is None:
            t = st["chwn"][0]
            st["nchw"] = ops.from_batch_innermost(t) if (t.shape[1] * t.shape[2] > 1 or not st.get("flat")) else t.reshape(t.shape[0], B).t().contiguous()
        return st["nchw"]

    def put(chwn_t=None, nchw_t=None):
        st["chwn"], st["nchw"] = chwn_t, nchw_t

    with ops.use_config(split_k=False):                      # (the per-layer path's promise: the plain chain, the reference-layout kernel's bits)
        for m in mods:
            hooked = bool(m._forward_hooks or m._forward_pre_hooks)
            inp = None
            if hooked:
                inp = nchw()
                for hook in list(m._forward_pre_hooks.values()):
                    r = hook(m, (inp,))
                    if r is not None:
                        inp = r[0] if isinstance(r, tuple) else r
                        if not (torch.is_tensor(inp) and inp.is_cuda and inp.dtype == torch.float32 and inp.shape[0] == B):
                            raise RuntimeError("a forward pre-hook replaced the input with something the batched path cannot run")
                        put(nchw_t=inp)
            if isinstance(m, BayesianLayer):
                is_conv = hasattr(m, "kernel_size")
                h = chwn()
                geom = (m.stride, m.padding, m.dilation) if is_conv else (1, 0, 1)
                if not is_conv:
                    h = h.reshape(1, m.in_features, 1, 1, B)
                if isinstance(m, _BBBLayer):
                    w, b = m._weights(True)
                    w5 = w if is_conv else w.reshape(1, m.out_features, m.in_features, 1, 1)
                    y = ops.conv2d_chwn_forward(h, w5, b, *geom, bf16x3=False)
                else:
                    w_var, b_var = m._variances()
                    seed, call = rng.layer_call()
                    w_mu = m.W_mu if is_conv else m.W_mu.reshape(m.out_features, m.in_features, 1, 1)
                    if not is_conv:
                        w_var = w_var.reshape(m.out_features, m.in_features, 1, 1)
                    y = ops.lrt_conv2d_chwn_forward(h, w_mu, w_var, m.bias_mu if m.use_bias else None, b_var, seed, call, m._stream_base + 2,
                                                    *geom, sample=True)[0]
                st["flat"] = not is_conv
                put(chwn_t=y)
            elif isinstance(m, nn.ReLU):
                t = st["chwn"] if st["chwn"] is not None else st["nchw"]
                r = torch.relu(t)
                put(chwn_t=r) if st["chwn"] is not None else put(nchw_t=r)
            elif isinstance(m, nn.Softplus):
                t = st["chwn"] if st["chwn"] is not None else st["nchw"]
                r = F.softplus(t, m.beta, m.threshold)
                put(chwn_t=r) if st["chwn"] is not None else put(nchw_t=r)
            elif isinstance(m, nn.MaxPool2d):
                put(chwn_t=ops.maxpool_chwn(chwn(), m.kernel_size, m.stride if m.stride is not None else m.kernel_size))
            elif isinstance(m, FlattenLayer):
                h = chwn()
                if h.shape[1] * h.shape[2] * h.shape[3] != m.num_features:
                    raise RuntimeError("hooked_chain: a flatten that changes the row count slipped through the checks")
                st["flat"] = True
                put(chwn_t=h.reshape(1, m.num_features, 1, 1, B))
            else:
                raise RuntimeError("hooked_chain: module %s slipped through the checks" % type(m).__name__)
            if hooked:
                out = nchw()
                for hook in list(m._forward_hooks.values()):
                    r = hook(m, (inp,), out)
                    if r is not None:
                        out = r
                        if not (torch.is_tensor(out) and out.is_cuda and out.dtype == torch.float32 and out.shape[0] == B):
                            raise RuntimeError("a forward hook replaced the output with something the batched path cannot run")
                        put(nchw_t=out)
    return nchw()


def fast_forward(wrapper, x):
    """Whole-model forward on the batch-innermost path of bbb_hip.ensemble (pixel-major GEMMs that skip padding
    taps, activation fused into the epilogue, HIP pooling): the same kernels -- hence the same bits -- as draw j
    of a batched mc_forward under the same call index.  Applies when the input is a CUDA [B, C, H, W] batch with B % 4 == 0
    that does not itself require a gradient, the model is made of modules that path knows, it ends in a Bayesian linear layer,
    no layer replays external noise, no module of it carries forward hooks (the fused path does not call the children), and
    this is the outermost wrapper of the forward.  With autograd enabled the same kernels run behind ONE autograd node
    (bbb_hip.fast_train).  Consecutive calls on the same input are batched speculatively (_Spec).
    Returns (output, kl) or None (-> the reference-layout path below, which also serves input gradients and hooks)."""
    from ._base import BayesianLayer
    if isinstance(wrapper, BayesianLayer) or rng._scope or not torch.is_tensor(x) or not x.is_cuda or x.dim() != 4:
        return None
    if x.requires_grad and torch.is_grad_enabled():
        return None                                   # saliency / adversarial inputs: the reference-layout path has d/dx
    from bbb_hip import ensemble
    mods = ensemble.flat_children(wrapper)
    if not mods or not isinstance(mods[-1], BayesianLayer) or not hasattr(mods[-1], "out_features"):
        return None
    layers = ensemble.bayesian_layers(wrapper)
    if any(l.eps_source is not None for l in layers) or not all(l.W_mu.is_cuda for l in layers):
        return None
    if _hooks_present(wrapper, mods):
        return None
    grad = torch.is_grad_enabled() and ensemble.any_requires_grad(wrapper)
    from bbb_hip import ops
    prec = ops.current_config().dropin_precision
    if prec != "fp32":
        if grad:
            return None                               # (ModuleWrapper.forward raises: bf16 is inference only)
        ensemble._check_precision(prec, wrapper, x, True)     # raises unless the bf16 inference path covers this model and input
    if grad:
        # training / the reference's validate loop (which does not disable autograd): the same kernels behind ONE autograd node
        from bbb_hip import fast_train
        if not ensemble.fast_autograd or not fast_train.train_path_ok(wrapper, x):
            return None
    elif not ensemble._chwn_ok(wrapper, x):
        return None

    # ---- speculative draw batching ----
    sp = wrapper.__dict__.get("_bbb_spec")
    if sp is None:
        sp = wrapper.__dict__["_bbb_spec"] = _Spec()
    st = ensemble._structure(wrapper)
    # parameter versions AND storage addresses (module.to() / .data assignments replace storage without a version bump), plus the
    # identity of the cached structure (a replaced layer or Parameter object rebuilds it)
    pver = (id(st),) + tuple((p._version, p.data_ptr()) for p in st["params"])
    mode = (grad, prec)                               # (cached draws belong to one autograd mode and one precision)
    same_x = sp.xref is not None and sp.xref() is x and sp.xver == x._version and sp.pver == pver and sp.grad == mode
    can_spec = speculation["enabled"] and rng._graph_counter["tensor"] is None and not torch.cuda.is_current_stream_capturing()
    seed, call = rng.get_state()
    if same_x and grad and sp.logits is not None and getattr(sp.logits, "bbb_cfg", {}).get("spent"):
        sp.logits = None                              # that graph's backward has run: its remaining draws cannot be used
    if same_x and can_spec and sp.logits is not None and sp.idx < sp.logits.shape[0] and sp.seed == seed and sp.next_call == call:
        rng.next_calls(1)                             # the generator moves exactly as the loop would move it
        j = sp.idx
        sp.idx += 1
        sp.next_call = (call + 1) & 0xFFFFFFFF
        sp.streak += 1
        for l in layers:
            l.__dict__["_kl"] = None
        return (sp.logits[j].t() if grad else sp.logits[j]), sp.kl
    if same_x:
        sp.streak += 1
    else:
        if sp.streak:
            sp.last_streak = sp.streak
        sp.streak, sp.batches = 1, 0
    K = 1
    if can_spec:
        done = sp.streak - 1                          # calls of this streak already answered
        if sp.last_streak > done + 1:
            K = sp.last_streak - done                 # history: the loop makes last_streak calls per input
        elif done >= 1:
            K = 2 << min(sp.batches, 8)               # no (or exhausted) history: 2, 4, 8, ...
        K = max(1, min(K, int(speculation["max_draws"])))
    seed, call = rng.next_calls(1)
    if grad:
        from bbb_hip import fast_train
        logits, kl = fast_train.mc_logits_autograd(wrapper, x, K, seed, call)
    else:
        g = _logit_graph(wrapper, st, x, K, seed, pver, prec) if K > 1 else None
        if g is not None:
            logits, kl = g.run(x, call)               # one graph launch; the buffers are the graph's: keep our own copy
            logits, kl = logits.clone(), kl.clone()
        else:
            out = ensemble._mc_logits_chwn(wrapper, x, K, seed, call, precision=prec)
            if out is None:
                rng.rewind((seed, call))              # nothing was launched: give the call index back
                return None
            logits, kl = out[0].permute(0, 2, 1).contiguous(), out[1]   # [K, B', C]: ONE transpose per batch of draws, not one per call
    for l in layers:
        l.__dict__["_kl"] = None
    import weakref
    sp.xref, sp.xver, sp.pver, sp.grad = weakref.ref(x), x._version, pver, mode
    if K > 1:
        sp.seed, sp.next_call, sp.logits, sp.kl, sp.idx = seed, (call + 1) & 0xFFFFFFFF, logits, kl, 1
        sp.batches += 1
    else:
        sp.logits = None
    return (logits[0].t() if grad else logits[0]), kl
