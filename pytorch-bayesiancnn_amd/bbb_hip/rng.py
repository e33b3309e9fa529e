"""Seed / call-counter contract of the on-chip noise (see include/bbb_hip.h, "Noise contract").

The reference draws eps from torch's default CPU generator.  Here every stochastic forward consumes
one *call index*; Monte-Carlo draw j of a batched E-draw launch uses call0 + j, so `for j in range(E): net(x)` and one
batched E-draw launch see identical noise.

Where (seed, call) come from -- SURVEY.md section 8(b): "seed + offset from torch's HIP generator":
  * default: torch's default generator of the CURRENT HIP DEVICE.  seed = its initial_seed(); call = its Philox offset / 4; a
    forward that needs n call indices advances the offset by 4n, exactly as a torch kernel consuming n Philox counters would.
    So `torch.manual_seed` / `torch.cuda.manual_seed(_all)` reseed the noise, `torch.cuda.get_rng_state` / `set_rng_state`
    save and replay it, and torch's own random kernels interleave with it in one stream of offsets.
  * pinned: after `rng.manual_seed(seed, call)` a private counter is used instead and torch's generators are left alone
    (tests, and jobs that want noise independent of whatever else draws random numbers).  Reseeding torch un-pins.
  * no HIP device (the CPU test suite) or while a hipGraph is being captured (generator state must not move inside a capture):
    the private counter, keyed on torch.initial_seed().
All ranks of a multi-GPU job must hold the same (seed, call) and advance it in lock step -- same torch seed and the same
sequence of random draws on every rank (bench.py), or a pinned stream -- which is what lets any rank materialise any draw
without communication.
"""
import torch

_state = {"seed": None, "call": 0, "torch_seed": None, "pinned": False}


def _device_generator():
    """The default generator of the current HIP device, or None (no device / inside a graph capture / API missing)."""
    if not torch.cuda.is_available():
        return None
    try:
        if torch.cuda.is_current_stream_capturing():
            return None
        g = torch.cuda.default_generators[torch.cuda.current_device()]
        g.get_offset()
        return g
    except (RuntimeError, AttributeError, IndexError):
        return None


def _sync():
    ts = torch.initial_seed()
    if _state["torch_seed"] != ts:
        _state["torch_seed"] = ts
        _state["seed"] = ts & 0xFFFFFFFFFFFFFFFF
        _state["call"] = 0
        _state["pinned"] = False


def manual_seed(seed, call=0):
    """Pin the noise stream explicitly (does not touch torch's generators)."""
    _state["torch_seed"] = torch.initial_seed()
    _state["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _state["call"] = int(call)
    _state["pinned"] = True


def use_device_generator():
    """Undo manual_seed: (seed, call) come from torch's default generator of the current device again."""
    _state["pinned"] = False
    _state["torch_seed"] = None


def get_state():
    _sync()
    g = None if _state["pinned"] else _device_generator()
    if g is not None:
        return g.initial_seed() & 0xFFFFFFFFFFFFFFFF, (g.get_offset() // 4) & 0xFFFFFFFF
    return _state["seed"], _state["call"]


def next_calls(n=1):
    """Reserve n consecutive call indices; returns (seed, first_call)."""
    _sync()
    g = None if _state["pinned"] else _device_generator()
    if g is not None:
        off = g.get_offset()
        if n:
            g.set_offset(off + 4 * int(n))
        return g.initial_seed() & 0xFFFFFFFFFFFFFFFF, (off // 4) & 0xFFFFFFFF
    c = _state["call"]
    _state["call"] = (c + int(n)) & 0xFFFFFFFF
    return _state["seed"], c


def rewind(seed_call):
    """Give back the call indices reserved since `seed_call` = (seed, call) was returned by next_calls (nothing was launched)."""
    g = None if _state["pinned"] else _device_generator()
    if g is not None:
        g.set_offset(4 * int(seed_call[1]))
    else:
        _state["call"] = int(seed_call[1])


_scope = []   # (seed, call) of the whole-model forward in progress (ModuleWrapper.forward), innermost last


def push_forward_scope():
    """Called by a model-level ModuleWrapper.forward: ONE call index for every layer of this forward, so that
    `net(x)` number j of a Python MC loop == draw j of a batched launch."""
    sc = next_calls(1)
    _scope.append(sc)
    return sc


def pop_forward_scope():
    _scope.pop()


def layer_call():
    """Call index a layer's forward should use: the enclosing model forward's, or a fresh one if the layer is
    used stand-alone."""
    if _scope:
        return _scope[-1]
    return next_calls(1)


_next_stream = [0]


def new_stream_base():
    """Four stream ids per Bayesian layer: +0 weight, +1 bias, +2 activation (LRT)."""
    b = _next_stream[0]
    _next_stream[0] += 4
    return b


def assign_stream_ids(net):
    """Number the Bayesian layers of `net` in module order (0, 4, 8, ...), so the noise of a model does
    not depend on what else was constructed in the process (needed for rank-to-rank agreement)."""
    i = 0
    for m in net.modules():
        if hasattr(m, "_stream_base"):
            m._stream_base = 4 * i
            i += 1
    return i


# ---- device-side call offset (hipGraph replay) ---------------------------------------------------------------
_graph_counter = {"tensor": None}


def call_dev_ptr(device):
    """Pointer handed to the kernels as `call_dev`: 0 (NULL) in eager mode; while a GraphedMC step is being captured or
    exists, the address of its device counter so that every replay draws fresh noise."""
    t = _graph_counter["tensor"]
    return 0 if t is None else t.data_ptr()


class device_call_offset:
    """Context: kernels launched inside add `counter` (a 1-element int32 device tensor) to their call index."""

    def __init__(self, counter):
        self.counter = counter

    def __enter__(self):
        self.prev = _graph_counter["tensor"]
        _graph_counter["tensor"] = self.counter
        return self

    def __exit__(self, *exc):
        _graph_counter["tensor"] = self.prev
        return False
