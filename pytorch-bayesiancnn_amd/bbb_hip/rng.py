"""Seed / call-counter contract of the on-chip noise (see include/bbb_hip.h, "Noise contract").

The reference draws eps from torch's default CPU generator and never seeds it.  Here
every stochastic forward consumes one *call index*; Monte-Carlo draw j of a batched E-draw launch uses
call0 + j, so `for j in range(E): net(x)` and one batched E-draw launch see identical noise.  The seed
follows ``torch.manual_seed``: whenever torch's initial seed changes, the call counter restarts at 0.
All ranks of a multi-GPU job hold the same (seed, call) and advance it in lock step, which is what lets
any rank materialise any draw without communication.
"""
import torch

_state = {"seed": None, "call": 0, "torch_seed": None}


def _sync():
    ts = torch.initial_seed()
    if _state["torch_seed"] != ts:
        _state["torch_seed"] = ts
        _state["seed"] = ts & 0xFFFFFFFFFFFFFFFF
        _state["call"] = 0


def manual_seed(seed, call=0):
    """Pin the noise stream explicitly (does not touch torch's generators)."""
    _state["torch_seed"] = torch.initial_seed()
    _state["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _state["call"] = int(call)


def get_state():
    _sync()
    return _state["seed"], _state["call"]


def next_calls(n=1):
    """Reserve n consecutive call indices; returns (seed, first_call)."""
    _sync()
    c = _state["call"]
    _state["call"] = (c + int(n)) & 0xFFFFFFFF
    return _state["seed"], c


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
