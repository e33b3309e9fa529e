"""Training extension: the inner loop of the reference's train_model
(main_bayesian.py:36-62) on the MI355X path -- Monte-Carlo forward with autograd through the HIP kernels
(bbb_hip.ops), the ELBO of metrics.py:12-24, one multi-tensor Adam launch, and, for data-parallel jobs (one process
per GPU, each with its own shard of the batch), ONE all-reduce of the flattened gradients per step.

Not part of the forward metric; forward-only inference is `bbb_hip.ensemble`.
"""
import weakref

import torch
import torch.nn.functional as F

from . import _lib, ensemble, ops, rng
from ._lib import AdamSegment, check, cur_stream


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) semantics (amsgrad off, no weight decay) with the update of up to 16
    parameter tensors per HIP launch (`bbb_adam_step`).  State layout matches torch's ('step', 'exp_avg', 'exp_avg_sq'),
    so state_dicts are interchangeable with torch.optim.Adam.
    capturable=True keeps 'step' on the device (torch's capturable convention) and lets the kernel derive the bias
    corrections from it, so the step can be part of a captured hipGraph (GraphedTrainStep).  The learning rate of a
    capturable group also lives on the device (a private per-group scalar, kept OUT of param_groups so that state_dict() holds
    exactly what torch.optim.Adam's does): `sync_lr()` copies group['lr'] there, which is how a scheduler's change reaches a
    step that was captured earlier."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, capturable=bool(capturable)))
        self._lr_dev = {}          # group index -> [device scalar, last value pushed]

    def make_capturable(self):
        """Switch every group to the capturable convention in place: 'step' counts move to the device (same values)."""
        for group in self.param_groups:
            if group.get("capturable", False):
                continue
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st and not st["step"].is_cuda:
                    st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
            group["capturable"] = True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # new 'step' / 'exp_avg' / 'exp_avg_sq' tensors replaced the ones an earlier capture (train_step's self-capture,
        # GraphedTrainStep) baked in: the generation is part of train_step's capture key, so such a graph is dropped
        self._state_generation = getattr(self, "_state_generation", 0) + 1
        for rec in self._lr_dev.values():     # the captured step keeps reading OUR scalars: refresh them from the loaded lr
            rec[1] = None
        if self._lr_dev and not torch.cuda.is_current_stream_capturing():
            self.sync_lr()

    def sync_lr(self):
        """Push every capturable group's current lr to its device scalar (call outside a graph capture, e.g. after
        scheduler.step(); GraphedTrainStep.step does it before each replay)."""
        for gi, group in enumerate(self.param_groups):
            rec = self._lr_dev.get(gi)
            if rec is not None and rec[1] != float(group["lr"]):
                rec[0].fill_(float(group["lr"]))
                rec[1] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        touched = []
        for gi, group in enumerate(self.param_groups):
            cap = group.get("capturable", False)
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_device(p, p.grad)
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.BBBHipError("FusedAdam needs contiguous parameters and gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if cap else torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if cap:
                    by_step.setdefault(0, []).append((p, st))
                else:
                    st["step"] += 1
                    by_step.setdefault(int(st["step"].item()), []).append((p, st))
            lr_dev = 0
            if cap and by_step:
                steps = [st["step"] for _, st in by_step[0]]
                if any(not t.is_cuda for t in steps):
                    raise _lib.BBBHipError("capturable FusedAdam needs its 'step' state on the device")
                torch._foreach_add_(steps, 1.0)              # one launch; all tensors of a group step together
                # the device-side learning rate lives in a private dict keyed by group index, NOT in param_groups: state_dict()
                # stays interchangeable with torch.optim.Adam's, and load_state_dict / map_location cannot swap or move the
                # scalar a captured step reads
                if gi not in self._lr_dev:
                    self._lr_dev[gi] = [torch.full((), float(group["lr"]), dtype=torch.float32, device=steps[0].device), float(group["lr"])]
                elif not torch.cuda.is_current_stream_capturing():
                    self.sync_lr()
                lr_dev = self._lr_dev[gi][0].data_ptr()
            for step, items in by_step.items():
                for s0 in range(0, len(items), _lib.MAX_SEGMENTS):
                    part = items[s0:s0 + _lib.MAX_SEGMENTS]
                    segs = (AdamSegment * len(part))()
                    for i, (p, st) in enumerate(part):
                        segs[i].param, segs[i].grad = p.data_ptr(), p.grad.data_ptr()
                        segs[i].exp_avg, segs[i].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        segs[i].n = p.numel()
                    dev = part[0][0].device
                    with torch.cuda.device(dev):
                        check(L.bbb_adam_step(segs, len(part), float(group["lr"]), float(group["betas"][0]),
                                              float(group["betas"][1]), float(group["eps"]), step,
                                              part[0][1]["step"].data_ptr() if cap else 0, lr_dev, cur_stream(dev)),
                              "bbb_adam_step")
                    touched += [p for p, _ in part]
        _bump_versions(touched)
        return loss


def _bump_versions(params):
    """The HIP kernels write parameters through raw pointers: tell torch (Tensor._version) so that version-keyed caches --
    the split-fp16 weight bound, the speculation cache of layers/_fused.py, fast_train's stale-parameter check -- see the
    update.  Host-side bookkeeping only (legal inside a capture; replays are bumped by GraphedTrainStep.step)."""
    if params:
        torch.autograd.graph.increment_version(params)


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20):
    """Data-parallel gradient averaging: gradients are packed into flat fp32 buckets (one for a BayesianAlexNet: 20 MB)
    and each bucket is ONE all_reduce (RCCL on GPUs); sum / world, written back in place.  Every rank must call this
    with the same parameter order.  Returns the number of collectives issued."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return 0
    n_coll, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * 4 <= bucket_bytes):
            size += grads[j].numel() * 4
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
        i = j
    return n_coll


def elbo(log_outputs, target, kl, beta, train_size):
    """metrics.ELBO.forward (metrics.py:7-14): nll_loss(mean) * train_size + beta * kl."""
    return F.nll_loss(log_outputs, target, reduction="mean") * train_size + beta * kl


def forward_loss(net, x, target, num_ens, beta, train_size, seed_call=None, param_alias=None):
    """The forward half of a training iteration (main_bayesian.py:43-56): num_ens stochastic forwards batched over the draws,
    kl of one forward (= the reference's kl / num_ens), logmeanexp, ELBO -> (loss, log_outputs, kl).  On the batch-innermost
    autograd path the loss tail is two HIP launches forward and one backward (ops.elbo_cb_autograd); elsewhere torch ops.
    seed_call: (seed, call0) reserved by the caller (a captured step); default: num_ens fresh call indices."""
    if seed_call is None:
        rng.assign_stream_ids(net)
        seed_call = rng.next_calls(int(num_ens))
    req = [target, beta, float(train_size), None]
    log_outputs, kl = ensemble._local_lse(net, x, int(num_ens), seed_call[0], seed_call[1], int(num_ens), param_alias=param_alias, elbo=req)
    loss = req[3] if req[3] is not None else elbo(log_outputs, target, kl, beta, train_size)
    return loss, log_outputs, kl


auto_graph = {"enabled": True, "after": 3,       # train_step captures itself once this many identical calls in a row were seen,
              "max_rows": 2048}                  # for steps of fewer than this many (image x draw) rows: larger steps are GPU-bound, and launch by
                                                 # launch their weight gradients overlap the input gradients on a second stream, which a
                                                 # captured step cannot do profitably (bs 512 x 10: 2.83 ms eager, 3.04 captured)



class _AutoState(dict):
    """Per-net self-capture state; copies and pickles of the net do not take it along (a captured graph belongs to one module)."""

    def __deepcopy__(self, memo):
        return _AutoState()

    def __reduce__(self):
        return (_AutoState, ())


class _AutoStates:
    """net -> {"key", "streak", "graphed"}, kept ON the module (net.__dict__): the captured step references its net, so a
    WeakKeyDictionary entry would keep its own key alive for ever; a net -> state -> step -> net cycle is ordinary garbage."""

    @staticmethod
    def get(net, default=None):
        return net.__dict__.get("_bbb_train_auto") or default        # (an empty state = a copied net's placeholder)

    def __getitem__(self, net):
        return net.__dict__["_bbb_train_auto"]

    def __setitem__(self, net, st):
        net.__dict__["_bbb_train_auto"] = _AutoState(st)

    @staticmethod
    def pop(net, default=None):
        return net.__dict__.pop("_bbb_train_auto", default)


_auto = _AutoStates()


def _python_hooks(net, optimizer):
    """Anything a graph replay would silently skip: module / tensor / optimizer hooks."""
    import torch.nn.modules.module as _m
    if (_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks or _m._global_backward_pre_hooks
            or optimizer._optimizer_step_pre_hooks or optimizer._optimizer_step_post_hooks):
        return True
    for m in net.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return True
    return any(getattr(p, "_backward_hooks", None) for p in net.parameters())


def _auto_key(net, optimizer, x, target, num_ens, train_size):
    """What a captured step bakes in; None = this call cannot be a graph replay (-> eager)."""
    if not (auto_graph["enabled"] and isinstance(optimizer, FusedAdam) and torch.is_tensor(x) and x.is_cuda and target.is_cuda
            and torch.is_grad_enabled() and not x.requires_grad and not torch.cuda.is_current_stream_capturing()):
        return None
    # parameter STORAGE and optimizer-state identity are baked into a captured step too: net.to(...), `p.data = ...` or
    # optimizer.load_state_dict(...) after the capture must drop it (the replay would read and write freed buffers)
    groups = tuple((tuple(g["betas"]), float(g["eps"]), tuple((id(p), p.requires_grad, p.data_ptr()) for p in g["params"]))
                   for g in optimizer.param_groups)
    from . import rng
    return (id(optimizer), getattr(optimizer, "_state_generation", 0), tuple(x.shape), x.dtype, tuple(target.shape), target.dtype,
            int(num_ens), float(train_size), net.training,
            ensemble._epoch[0], groups, rng.next_calls(0)[0])               # the noise seed is a constant of the captured kernels


def train_step(net, optimizer, x, target, num_ens, beta, train_size, dp_group=None, graph=None):
    """One iteration of train_model's batch loop (main_bayesian.py:40-58): zero_grad, num_ens stochastic forwards
    (batched over draws), kl / num_ens, logmeanexp, ELBO, backward, [gradient all-reduce], optimizer.step.
    dp_group: data-parallel process group whose ranks hold different shards of the batch (parameters replicated).
    Returns (loss, log_outputs, kl) detached.

    A fixed-shape loop is host-bound when run launch by launch (~40 launches + autograd bookkeeping per iteration), so by default
    (graph=None) the step captures ITSELF: once the same (optimizer, shapes, num_ens, train_size, parameter set) was seen
    auto_graph["after"] times in a row, single-process, with a FusedAdam and no Python hooks on the model, the call runs one more
    eager iteration on a capture stream, records it as a GraphedTrainStep and replays that from then on (beta and the learning
    rate stay run-time values; a non-capturable FusedAdam is switched to capturable in place).  Any change of the key drops the
    graph and returns to launch-by-launch steps.  graph=False: never capture.  Same noise calls, same results as the eager
    sequence (tests/test_gpu_train.py)."""
    small = graph is True or x.shape[0] * int(num_ens) < int(auto_graph.get("max_rows", 1 << 62))     # (graph=True: capture whatever the size)
    key = _auto_key(net, optimizer, x, target, num_ens, train_size) if (graph is not False and dp_group is None and small) else None
    st = _auto.get(net)
    if key is None or st is None or st["key"] != key:
        if key is not None:
            _auto[net] = {"key": key, "streak": 0, "graphed": None}
            st = _auto[net]
        else:
            _auto.pop(net, None)
    if key is not None:
        if st["graphed"] is not None:
            loss, log_outputs, kl = st["graphed"].step(x, target, beta)
            return loss.clone(), log_outputs.clone(), kl.clone()          # the graph's outputs are overwritten by the next replay
        st["streak"] += 1
        if st["streak"] > int(auto_graph["after"]) and not _python_hooks(net, optimizer):
            optimizer.make_capturable()
            g = GraphedTrainStep(net, optimizer, x, target, num_ens, beta, train_size, warmup=1, weak_net=True)   # the warm-up IS this iteration
            if g.graph is not None:
                st["graphed"] = g
            else:
                st["streak"] = -(1 << 60)            # capture refused (stale gradient accumulators): launch by launch from here on
            return g.warm_loss.clone(), g.warm_log_outputs.clone(), g.warm_kl.clone()
    optimizer.zero_grad()
    loss, log_outputs, kl = forward_loss(net, x, target, num_ens, beta, train_size)
    loss.backward()
    if dp_group is not None:
        allreduce_gradients([p for g in optimizer.param_groups for p in g["params"]], dp_group)
    optimizer.step()
    return loss.detach(), log_outputs.detach(), kl.detach()


class GraphedTrainStep:
    """train_step captured as ONE hipGraph: Monte-Carlo forward with autograd, ELBO, backward, noise-counter increment
    and the Adam update replay as a single graph launch (the eager step is host-bound: ~40 kernel launches plus autograd
    bookkeeping per iteration).  Fresh noise per replay through the same device-side call counter as GraphedMC (the
    backward kernel reads it too, so it regenerates the forward's eps); the Adam step count lives on the device.
    Fixed shapes: copy each batch into `self.x` / `self.target` (or pass them to step()).  train_size is a constant of the
    captured graph; the KL weight `beta` (a per-batch schedule in the reference: metrics.get_beta, main_bayesian.py:55) and
    the learning rate (ReduceLROnPlateau, main_bayesian.py:118) are DEVICE scalars the graph reads at run time:
    step(beta=...) sets the former, optimizer.param_groups[i]['lr'] is pushed before every replay.
    `warmup` real training iterations run eagerly on the capture stream first.
    The optimizer must be FusedAdam(capturable=True) (or another capturable optimizer that keeps lr on the device)."""

    def __init__(self, net, optimizer, x, target, num_ens, beta, train_size, warmup=3, weak_net=False, launch_config=None):
        from . import rng
        _lib.require_device(x)
        self.launch_config = (launch_config if launch_config is not None else ops.current_config()).copy()
        # weak_net (train_step's self-capture keeps this object ON the net): no net -> state -> step -> net cycle, so dropping the
        # net frees the graph then and there instead of whenever the cyclic collector runs (possibly inside another capture)
        self._net = weakref.ref(net) if weak_net else (lambda n=net: n)
        self.opt, self.num_ens = optimizer, int(num_ens)
        self.train_size = float(train_size)
        self.x, self.target = x.clone(), target.clone()
        dev = x.device
        self.beta = torch.full((), float(beta), dtype=torch.float32, device=dev)
        self.counter = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.seed, self.call0 = rng.next_calls(0)
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        import warnings
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            with torch.cuda.stream(self.stream), rng.device_call_offset(self.counter):
                for _ in range(max(1, int(warmup))):
                    self.opt.zero_grad(set_to_none=True)
                    self.warm_loss, self.warm_log_outputs, self.warm_kl = self._body()      # results of the last eager iteration
                    rng.next_calls(self.num_ens)
        self._next_call = rng.next_calls(0)[1]
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        # A parameter's gradient accumulator from an earlier backward on ANOTHER stream, kept alive by a still-referenced
        # non-detached output of that forward, makes autograd synchronise with that stream -- fine in the warm-up iterations above
        # (torch warns), fatal inside a capture (observed: a segmentation fault in capture_end).  Refuse to capture instead.
        self.capture_refused = next((str(w.message).split(".")[0] for w in seen if "AccumulateGrad node's stream" in str(w.message)), None)
        for w in seen:
            if "AccumulateGrad node's stream" not in str(w.message):
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        self.graph = None
        self.replays = 0
        if self.capture_refused:
            return
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with rng.device_call_offset(self.counter), ops.graph_capture(self.graph, self.stream):
            self.loss, self.log_outputs, self.kl = self._body()

    @property
    def net(self):
        return self._net()

    def _body(self):
        with ops.use_config(self.launch_config), ops.scratch_scope(self):
            return self._body_inner()

    def _body_inner(self):
        from . import fast_train
        params = [p for g in self.opt.param_groups for p in g["params"] if p.requires_grad]
        # The autograd graph of a captured step is rooted at FRESH leaves that share the parameters' storage, and differentiated
        # with torch.autograd.grad: the parameters' own gradient accumulators never take part.  (One that an earlier backward
        # created on another stream -- kept alive by a still-referenced, non-detached output of that forward -- makes autograd
        # synchronise with that stream: harmless eagerly, a segmentation fault inside a capture.)  The batch-innermost training
        # path takes the leaves as arguments; on the reference-layout path the layers read their own parameters, and the
        # warm-up iteration's stream-mismatch warning is the (best-effort) signal to refuse the capture.
        alias = None
        if ensemble.fast_autograd and fast_train.train_path_ok(self.net, self.x):
            alias = {id(p): p.detach().requires_grad_(True) for p in params}
        loss, log_outputs, kl = forward_loss(self.net, self.x, self.target, self.num_ens, self.beta, self.train_size,   # kl of one forward
                                             seed_call=(self.seed, self.call0), param_alias=alias)           # = kl / num_ens of the sum
        leaves = params if alias is None else [alias[id(p)] for p in params]
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        for p, g in zip(params, grads):
            p.grad = g                               # (captured: the tensor lives in the graph's pool; replays rewrite it in place)
        self.counter.add_(self.num_ens)              # after backward: it regenerates eps from the same counter value
        self.opt.step()
        return loss.detach(), log_outputs.detach(), kl.detach()

    def step(self, x=None, target=None, beta=None):
        from . import rng
        if self.graph is None:
            raise _lib.BBBHipError("this step was not captured: " + str(self.capture_refused) + " (drop the references to the "
                                   "outputs of earlier non-detached forwards, then build the GraphedTrainStep again)")
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if beta is not None:
            self.beta.fill_(float(beta))
        if hasattr(self.opt, "sync_lr"):
            self.opt.sync_lr()
        seed, call = rng.next_calls(0)
        if call != self._next_call:                  # someone else drew noise in between (a validation pass): follow the host counter
            d = (call - self.call0) & 0xFFFFFFFF
            self.counter.fill_(d - (1 << 32) if d >= (1 << 31) else d)     # the kernels add it modulo 2^32
        self.graph.replay()
        self.replays += 1
        _bump_versions([p for g in self.opt.param_groups for p in g["params"] if p.requires_grad])   # the replay rewrote them
        rng.next_calls(self.num_ens)                 # keep the host-side noise counter in step with the device's
        self._next_call = call + self.num_ens
        return self.loss, self.log_outputs, self.kl
