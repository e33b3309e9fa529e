"""Archived from bbb_hip/ops.py (round 3): the host side of bbb_chain_fwd.  Not importable as is."""
# ---- one persistent launch for a whole step's layers (bbb_chain_fwd) ------------------------------------------------

def _chain_workspace(device, n):
    # one workspace per (device, stream): steps in flight on different streams (graph lanes) must not share counters
    key = (device.index, "chain", cur_stream(device))
    buf = _scratch.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(int(n), dtype=torch.int32, device=device)
        _scratch[key] = buf
    return buf


def chain_error(device):
    """Error word of the current stream's chain workspace (non-zero: a dependency wait timed out).  Synchronises."""
    buf = _scratch.get((torch.device(device).index, "chain", cur_stream(torch.device(device))))
    return 0 if buf is None else int(buf[8].item())


def chain_forward(x, specs, flags=0):
    """A chain of conv / linear / max-pool stages over `slabs` Monte-Carlo draws (or work units) as ONE persistent launch:
    the same numbers, bit for bit, as conv2d_chwn_forward / maxpool_chwn called stage by stage (include/bbb_hip.h,
    bbb_chain_fwd).  x: [E|1|S, Cin, H, W, B] input of the first stage (a conv).  specs, in order:
        ("conv", w, bias, stride, padding, dilation, act, out, kw)   w [E|.., Cout, Cin, kh, kw]; kw = units keywords of conv2d_chwn_forward
        ("pool", k, s)
        ("flatten", features)        free view [E, C, H, W, B] -> [E, features, 1, 1, B]
    Returns the last stage's output, or None when the geometry is outside what the chain kernel takes (caller falls back)."""
    require_device(x)
    cur = x.contiguous()
    stages = []
    keep = []
    prev = -1
    slabs = None
    for sp in specs:
        if sp[0] == "flatten":
            if prev < 0 or cur.shape[1] * cur.shape[2] * cur.shape[3] != sp[1]:
                return None
            cur = cur.reshape(cur.shape[0], sp[1], 1, 1, cur.shape[4])
            continue
        st = _lib.ChainStage()
        if sp[0] == "conv":
            _, w, bias, stride, padding, dilation, act, out, kw = sp
            require_device(w, bias)
            w = w.contiguous()
            bias = None if bias is None else bias.contiguous()
            units, n_units, x_per_slice = kw.get("units"), kw.get("n_units"), kw.get("x_per_slice", False)
            if units is not None and units[0] > 1:
                E = int(n_units)
                d, ho, wo = _desc_chwn(cur, w, stride, padding, dilation, E, False, False, act)
                _apply_units(d, units, x_per_slice)
            else:
                E = max(cur.shape[0], w.shape[0])
                if cur.shape[0] not in (1, E) or w.shape[0] not in (1, E):
                    raise _lib.BBBHipError("leading (draw) dims of x and w must be 1 or equal")
                d, ho, wo = _desc_chwn(cur, w, stride, padding, dilation, E, cur.shape[0] == 1 and E > 1, w.shape[0] == 1 and E > 1, act)
            shape = (E, w.shape[1], ho, wo, cur.shape[4])
            if out is None:
                y = torch.empty(shape, dtype=torch.float32, device=cur.device)
            else:
                if out.numel() != E * w.shape[1] * ho * wo * cur.shape[4] or not out.is_contiguous() or out.dtype != torch.float32:
                    raise _lib.BBBHipError("out= must be a contiguous fp32 tensor of the output's size")
                y = out.view(shape)
            st.kind, st.conv = _lib.CHAIN_CONV, d
            st.w, st.bias = w.data_ptr(), ptr(bias)
            keep += [w, bias]
        else:
            _, k, s_ = sp
            if prev < 0:
                return None                               # a pool on the raw input has no per-draw slabs
            E, C, H, W, B = cur.shape
            d = ConvDesc()
            d.batch, d.cin, d.h, d.w, d.kh, d.stride_h, d.draws = B, C, H, W, int(k), int(s_), E
            if H < k or W < k:
                return None
            y = torch.empty((E, C, (H - k) // s_ + 1, (W - k) // s_ + 1, B), dtype=torch.float32, device=cur.device)
            st.kind, st.conv = _lib.CHAIN_MAXPOOL, d
        if slabs is None:
            slabs = E
        elif E != slabs:
            return None
        st.dep = prev
        st.x, st.y = cur.data_ptr(), y.data_ptr()
        stages.append(st)
        keep += [cur, y]
        prev = len(stages) - 1
        cur = y
    if not stages or len(stages) > _lib.CHAIN_MAX_STAGES:
        return None
    arr = (_lib.ChainStage * len(stages))(*stages)
    L = _lib.lib()
    dev = cur.device
    n = L.bbb_chain_workspace(len(stages), slabs)
    if n <= 0:
        return None                                       # more slabs than the chain kernel schedules: per-layer launches
    with on_device(dev):
        ws = _chain_workspace(dev, n)
        rc = L.bbb_chain_fwd(arr, len(stages), int(flags), ws.data_ptr(), ws.numel(), cur_stream(dev))
    if rc in (-2, -3):                                    # alignment / geometry outside the chain kernel: per-layer launches
        return None
    check(rc, "bbb_chain_fwd")
    return cur




# ---- archived from bbb_hip/ensemble.py::_mc_logits_chwn (nested function) ----
    def run_chain():
        """All conv / linear / pool stages of the step as ONE persistent launch (bbb_chain_fwd), or None: not this shape."""
        specs, i, pending_flat = [], 0, None
        while i < len(children):
            mod = children[i]
            nxt = children[i + 1] if i + 1 < len(children) else None
            if isinstance(mod, _BBBLayer):
                is_conv = isinstance(mod, _BBBConv)
                act = _act_name(nxt) if nxt is not None else None
                w, b = sampled[mod]
                if not is_conv:
                    w = w.reshape(w.shape[0], mod.out_features, mod.in_features, 1, 1)
                    if not specs:
                        return None                              # a model that starts with a linear layer: per-layer path
                geom = (mod.stride, mod.padding, mod.dilation) if is_conv else (1, 0, 1)
                kw = dict(ukw, x_per_slice=not specs) if ukw else {}
                dst = logits_buf if (logits_buf is not None and i == last_bayes and not is_conv) else None
                specs.append(("conv", w, b, *geom, act, dst, kw))
                if act is not None:
                    i += 1
            elif isinstance(mod, FlattenLayer):
                specs.append(("flatten", mod.num_features))
            elif isinstance(mod, nn.MaxPool2d):
                specs.append(("pool", mod.kernel_size, mod.stride))
            else:
                return None                                      # a stand-alone activation etc.
            i += 1
        if not tail_is_last or logits_buf is None:
            return None
        y = ops.chain_forward(xt, specs, flags=chain_flags)
        if y is None:
            return None
        return y.reshape(E, -1, xt.shape[-1])

