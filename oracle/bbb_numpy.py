"""CPU ORACLE (test infrastructure only) -- numpy restatement of the Bayes-by-Backprop hot path.

This file is a *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  Nothing under ``pytorch-bayesiancnn_amd/`` imports it and the
product path never falls back to it.

Every function restates one step of the reference (paths relative to the upstream repo
kumar-shridhar/PyTorch-BayesianCNN) and cites the file:line it follows.  Arithmetic is float32
wherever the reference's is (it is float32 everywhere); contractions accumulate in float64 and
round once, so the oracle is a tighter target than any fp32 accumulation order.

Parity pin: the reference ships no golden vectors (its tests are shape-only and stale), so this
restatement is pinned against *outputs of the reference itself*, generated in the build container
by ``tests/golden/make_golden.py`` (imports the unmodified upstream modules from /root/reference)
and committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks it against them.

The noise source is new (the reference draws eps from torch's CPU mt19937; the device path uses
counter-based Philox4x32-7).  ``philox4x32`` / ``normal_eps`` define that stream exactly and
are pinned by the Random123 known-answer vectors (7 and 10 rounds).
"""
import numpy as np

F32 = np.float32
U32 = np.uint32
U64 = np.uint64

# ----------------------------------------------------------------------------------------------
# Philox4x32-R (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).  The noise contract uses
# R = 7 (the paper's minimum round count that passes BigCrush; 10 is Random123's default safety margin -- round 1 of
# this project used 10; each round is 6 % of the device's whole parameter pass).
# Not in the reference: replaces `torch.empty(shape).normal_(0, 1)` (layers/BBB/BBBConv.py:63,68,
# layers/BBB/BBBLinear.py:56,61, layers/BBB_LRT/BBBConv.py:78, layers/BBB_LRT/BBBLinear.py:70).
# ----------------------------------------------------------------------------------------------
PHILOX_M0 = 0xD2511F53
PHILOX_M1 = 0xCD9E8D57
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
PHILOX_ROUNDS = 7


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=PHILOX_ROUNDS):
    """Vectorised Philox4x32-`rounds`.  All inputs broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=U64) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    mask = U64(0xFFFFFFFF)
    for r in range(rounds):
        p0 = U64(PHILOX_M0) * c0
        p1 = U64(PHILOX_M1) * c2
        hi0, lo0 = p0 >> U64(32), p0 & mask
        hi1, lo1 = p1 >> U64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ U64(k0)), lo1, (hi0 ^ c3 ^ U64(k1)), lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return c0.astype(U32), c1.astype(U32), c2.astype(U32), c3.astype(U32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    return philox4x32(c0, c1, c2, c3, k0, k1, rounds=10)


def box_muller(xa, xb):
    """Two uint32 words -> two N(0,1) float32.  The top 23 bits of each word are a mantissa:
    u1 = 1 - (xa >> 9) * 2^-23 in (0, 1],  u2 = (xb >> 9) * 2^-23 in [0, 1);  z = sqrt(-2 ln u1) * {cos, sin}(2 pi u2)."""
    u1 = 1.0 - (xa >> U32(9)).astype(np.float64) * 2.0 ** -23
    u2 = (xb >> U32(9)).astype(np.float64) * 2.0 ** -23
    r = np.sqrt(-2.0 * np.log(u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(F32), (r * np.sin(2.0 * np.pi * u2)).astype(F32)


def normal_eps(seed, call, stream, n, start=0):
    """The device noise stream: eps[i] for element index i in [start, start+n).

    counter = (i>>2 low 32 bits, i>>34, stream, call); key = (seed low, seed high);
    the 4 outputs of one Philox call give elements 4g..4g+3 via two Box-Muller pairs.
    """
    idx = np.arange(start, start + n, dtype=np.uint64)
    grp = idx >> U64(2)
    g = np.unique(grp)
    x0, x1, x2, x3 = philox4x32(g & U64(0xFFFFFFFF), g >> U64(32), U32(stream & 0xFFFFFFFF),
                                   U32(call & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    z0, z1 = box_muller(x0, x1)
    z2, z3 = box_muller(x2, x3)
    z = np.stack([z0, z1, z2, z3], axis=1).reshape(-1)
    off = int(start - int(g[0]) * 4)
    return z[off:off + n]


# ----------------------------------------------------------------------------------------------
# Parameter-space pieces
# ----------------------------------------------------------------------------------------------
def sigma_from_rho(rho):
    """sigma = log1p(exp(rho)) in float32 -- layers/BBB/BBBConv.py:64,69; BBBLinear.py:57,62;
    layers/BBB_LRT/BBBConv.py:64,66; BBB_LRT/BBBLinear.py:58,60.  (Overflows to inf for rho>~88,
    exactly like the reference; the device path returns rho itself above 20, which is the same
    float32 value wherever the reference's is finite.)"""
    rho = np.asarray(rho, dtype=F32)
    with np.errstate(over="ignore"):
        return np.log1p(np.exp(rho)).astype(F32)


def reparam(mu, sigma, eps):
    """w = mu + eps * sigma -- layers/BBB/BBBConv.py:65,70; BBBLinear.py:58,63."""
    return (np.asarray(mu, F32) + np.asarray(eps, F32) * np.asarray(sigma, F32)).astype(F32)


def kl_elements(mu, sigma, prior_mu, prior_sigma):
    """Per-element KL term AS THE REFERENCE CALLS IT.

    metrics.py:27-29 defines calculate_kl(mu_q, sig_q, mu_p, sig_p) =
        0.5 * (2*log(sig_p/sig_q) - 1 + (sig_q/sig_p)^2 + ((mu_p-mu_q)/sig_p)^2).sum()
    and every layer calls it as KL_DIV(prior_mu, prior_sigma, W_mu, W_sigma)
    (layers/BBB/BBBConv.py:80-82 and the three siblings), i.e. q := prior scalars, p := posterior
    tensors.  float32 elementwise, same operation order as the torch expression."""
    mu = np.asarray(mu, F32)
    sigma = np.asarray(sigma, F32)
    sq = F32(prior_sigma)
    mq = F32(prior_mu)
    t = F32(2.0) * np.log(sigma / sq) - F32(1.0) + (sq / sigma) ** 2 + ((mu - mq) / sigma) ** 2
    return (F32(0.5) * t).astype(F32)


def kl_loss(mu, sigma, prior_mu, prior_sigma):
    """Sum of kl_elements; float64 accumulation (the reference sums in float32 -- the golden test
    bounds the difference at 2e-6 relative)."""
    return float(np.sum(kl_elements(mu, sigma, prior_mu, prior_sigma), dtype=np.float64))


def kl_grads(mu, rho, prior_mu, prior_sigma):
    """d KL / d mu and d KL / d rho of the swapped form (what autograd gives the reference)."""
    mu = np.asarray(mu, np.float64)
    rho = np.asarray(rho, np.float64)
    s = np.log1p(np.exp(rho))
    d = mu - prior_mu
    gmu = d / s ** 2
    gs = 1.0 / s - prior_sigma ** 2 / s ** 3 - d ** 2 / s ** 3
    return gmu, gs / (1.0 + np.exp(-rho))


# ----------------------------------------------------------------------------------------------
# Contractions (F.conv2d / F.linear restated as im2col + matmul, float64 accumulate)
# ----------------------------------------------------------------------------------------------
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def im2col(x, kh, kw, stride, padding, dilation):
    """x [B,C,H,W] -> (cols [B*Ho*Wo, C*kh*kw], Ho, Wo); k ordered (c, kh, kw) like a
    [Cout, Cin, kh, kw] weight flattened per output channel."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    B, C, H, W = x.shape
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    xp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=x.dtype)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    cols = np.empty((B, Ho, Wo, C, kh, kw), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, :, i, j] = xp[:, :, i * dh:i * dh + sh * (Ho - 1) + 1:sh,
                                        j * dw:j * dw + sw * (Wo - 1) + 1:sw].transpose(0, 2, 3, 1)
    return cols.reshape(B * Ho * Wo, C * kh * kw), Ho, Wo


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1):
    """F.conv2d(x, w, b, stride, padding, dilation, groups=1) -- layers/BBB/BBBConv.py:77."""
    x = np.asarray(x, F32)
    w = np.asarray(w, F32)
    B = x.shape[0]
    Cout, Cin, kh, kw = w.shape
    cols, Ho, Wo = im2col(x, kh, kw, stride, padding, dilation)
    y = cols.astype(np.float64) @ w.reshape(Cout, -1).astype(np.float64).T
    if b is not None:
        y = y + np.asarray(b, np.float64)[None, :]
    return y.reshape(B, Ho, Wo, Cout).transpose(0, 3, 1, 2).astype(F32)


def linear(x, w, b=None):
    """F.linear(x, w, b) -- layers/BBB/BBBLinear.py:70."""
    y = np.asarray(x, np.float64) @ np.asarray(w, np.float64).T
    if b is not None:
        y = y + np.asarray(b, np.float64)[None, :]
    return y.astype(F32)


# ----------------------------------------------------------------------------------------------
# Layer forwards
# ----------------------------------------------------------------------------------------------
def bbb_conv2d_forward(x, W_mu, W_rho, bias_mu, bias_rho, W_eps, bias_eps,
                       stride=1, padding=0, dilation=1):
    """layers/BBB/BBBConv.py:61-77 with the two noise tensors passed in (eps replay).
    Returns (y, W_sigma, bias_sigma)."""
    W_sigma = sigma_from_rho(W_rho)
    weight = reparam(W_mu, W_sigma, W_eps)
    bias = bias_sigma = None
    if bias_mu is not None:
        bias_sigma = sigma_from_rho(bias_rho)
        bias = reparam(bias_mu, bias_sigma, bias_eps)
    return conv2d(x, weight, bias, stride, padding, dilation), W_sigma, bias_sigma


def bbb_linear_forward(x, W_mu, W_rho, bias_mu, bias_rho, W_eps, bias_eps):
    """layers/BBB/BBBLinear.py:54-70."""
    W_sigma = sigma_from_rho(W_rho)
    weight = reparam(W_mu, W_sigma, W_eps)
    bias = bias_sigma = None
    if bias_mu is not None:
        bias_sigma = sigma_from_rho(bias_rho)
        bias = reparam(bias_mu, bias_sigma, bias_eps)
    return linear(x, weight, bias), W_sigma, bias_sigma


def lrt_moments_conv2d(x, W_mu, W_rho, bias_mu, bias_rho, stride=1, padding=0, dilation=1):
    """act_mu, act_var of layers/BBB_LRT/BBBConv.py:64-74 (act_var includes the 1e-16)."""
    x = np.asarray(x, F32)
    W_sigma = sigma_from_rho(W_rho)
    bias_var = None
    if bias_mu is not None:
        bias_var = sigma_from_rho(bias_rho) ** 2
    act_mu = conv2d(x, W_mu, bias_mu, stride, padding, dilation)
    act_var = (F32(1e-16) + conv2d(x * x, W_sigma * W_sigma, bias_var, stride, padding, dilation)).astype(F32)
    return act_mu, act_var


def lrt_moments_linear(x, W_mu, W_rho, bias_mu, bias_rho):
    """act_mu, act_var of layers/BBB_LRT/BBBLinear.py:58-66."""
    x = np.asarray(x, F32)
    W_sigma = sigma_from_rho(W_rho)
    bias_var = None
    if bias_mu is not None:
        bias_var = sigma_from_rho(bias_rho) ** 2
    act_mu = linear(x, W_mu, bias_mu)
    act_var = (F32(1e-16) + linear(x * x, W_sigma * W_sigma, bias_var)).astype(F32)
    return act_mu, act_var


def lrt_output(act_mu, act_var, eps):
    """act_mu + sqrt(act_var) * eps -- layers/BBB_LRT/BBBConv.py:75,79; BBBLinear.py:67,71."""
    return (act_mu + np.sqrt(act_var) * np.asarray(eps, F32)).astype(F32)


# ----------------------------------------------------------------------------------------------
# Stock torch modules the models put between Bayesian layers (models/BayesianModels/*.py)
# ----------------------------------------------------------------------------------------------
def softplus_act(x):
    """nn.Softplus(beta=1, threshold=20)."""
    x = np.asarray(x, F32)
    with np.errstate(over="ignore"):
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, F32(20))))).astype(F32)


def relu_act(x):
    return np.maximum(np.asarray(x, F32), F32(0))


def maxpool2d(x, k, s):
    """nn.MaxPool2d(kernel_size=k, stride=s), no padding, floor mode."""
    B, C, H, W = x.shape
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    out = np.full((B, C, Ho, Wo), -np.inf, dtype=F32)
    for i in range(k):
        for j in range(k):
            out = np.maximum(out, x[:, :, i:i + s * (Ho - 1) + 1:s, j:j + s * (Wo - 1) + 1:s])
    return out


# ----------------------------------------------------------------------------------------------
# The MC-ensemble tail and the loss (main_bayesian.py:43-56, 73-83)
# ----------------------------------------------------------------------------------------------
def log_softmax(z, axis=1):
    z = np.asarray(z, np.float64)
    m = z.max(axis=axis, keepdims=True)
    return (z - m - np.log(np.exp(z - m).sum(axis=axis, keepdims=True))).astype(F32)


def logmeanexp(x, axis):
    """utils.py:14-22: x_max + log(mean(exp(x - x_max), dim))."""
    x = np.asarray(x, np.float64)
    m = x.max(axis=axis, keepdims=True)
    return np.squeeze(m + np.log(np.mean(np.exp(x - m), axis=axis, keepdims=True)), axis=axis).astype(F32)


def mc_log_outputs(logits_ebc):
    """logits [E,B,C] -> log_outputs [B,C]: per-draw log_softmax stored as outputs[:, :, j]
    (main_bayesian.py:49,78) then logmeanexp over the ensemble dim (main_bayesian.py:53,80)."""
    ls = log_softmax(np.asarray(logits_ebc), axis=2)
    return logmeanexp(ls, axis=0)


def elbo(log_outputs, target, kl, beta, train_size):
    """metrics.py:12-14: nll_loss(input, target, 'mean') * train_size + beta * kl."""
    lo = np.asarray(log_outputs, np.float64)
    nll = -np.mean(lo[np.arange(lo.shape[0]), np.asarray(target)])
    return float(nll * train_size + beta * kl)


def get_beta(batch_idx, m, beta_type, epoch=None, num_epochs=None):
    """metrics.py:32-46."""
    if type(beta_type) is float:
        return beta_type
    if beta_type == "Blundell":
        return 2 ** (m - (batch_idx + 1)) / (2 ** m - 1)
    if beta_type == "Soenderby":
        if epoch is None or num_epochs is None:
            raise ValueError("Soenderby method requires both epoch and num_epochs to be passed.")
        return min(epoch / (num_epochs // 4), 1)
    if beta_type == "Standard":
        return 1 / m
    return 0


def uncertainty(logits_tbc, normalized=False):
    """Aleatoric / epistemic decomposition from T stochastic forwards, uncertainty_estimation.py:37-58 (per image: T
    rows of one batch) and :61-102 (per batch: T forwards).  logits [T, B, C] -> (pred, epistemic, aleatoric), each [B, C]:
      p_hat = softmax(logits)                       (normalized: softplus(logits) / sum softplus(logits))
      pred = mean_T logits;  p_bar = mean_T p_hat
      epistemic = diag((p_hat - p_bar)^T (p_hat - p_bar)) / T;   aleatoric = diag(diag(p_bar) - p_hat^T p_hat / T)"""
    z = np.asarray(logits_tbc, np.float64)
    if normalized:
        sp = np.where(z > 20, z, np.log1p(np.exp(np.minimum(z, 20))))
        p = sp / sp.sum(axis=2, keepdims=True)
    else:
        m = z.max(axis=2, keepdims=True)
        e = np.exp(z - m)
        p = e / e.sum(axis=2, keepdims=True)
    pbar = p.mean(axis=0)
    epi = ((p - pbar[None]) ** 2).mean(axis=0)
    ale = pbar - (p ** 2).mean(axis=0)
    return z.mean(axis=0).astype(F32), epi.astype(F32), ale.astype(F32)


# ----------------------------------------------------------------------------------------------
# Model topologies as data (facts read from models/BayesianModels/*.py; attribute order there is
# the graph because ModuleWrapper.forward walks children(), layers/misc.py:16-18).
# ("conv", name, cout, k, stride, pad) | ("act",) | ("pool", k, s) | ("flatten", n) | ("fc", name, out)
# ----------------------------------------------------------------------------------------------
TOPOLOGY = {
    "lenet": [("conv", "conv1", 6, 5, 1, 0), ("act",), ("pool", 2, 2),
              ("conv", "conv2", 16, 5, 1, 0), ("act",), ("pool", 2, 2),
              ("flatten", 5 * 5 * 16), ("fc", "fc1", 120), ("act",), ("fc", "fc2", 84), ("act",),
              ("fc", "fc3", None)],
    "alexnet": [("conv", "conv1", 64, 11, 4, 5), ("act",), ("pool", 2, 2),
                ("conv", "conv2", 192, 5, 1, 2), ("act",), ("pool", 2, 2),
                ("conv", "conv3", 384, 3, 1, 1), ("act",),
                ("conv", "conv4", 256, 3, 1, 1), ("act",),
                ("conv", "conv5", 128, 3, 1, 1), ("act",), ("pool", 2, 2),
                ("flatten", 1 * 1 * 128), ("fc", "classifier", None)],
    "3conv3fc": [("conv", "conv1", 32, 5, 1, 2), ("act",), ("pool", 3, 2),
                 ("conv", "conv2", 64, 5, 1, 2), ("act",), ("pool", 3, 2),
                 ("conv", "conv3", 128, 5, 1, 1), ("act",), ("pool", 3, 2),
                 ("flatten", 2 * 2 * 128), ("fc", "fc1", 1000), ("act",), ("fc", "fc2", 1000), ("act",),
                 ("fc", "fc3", None)],
}


def model_forward(net_type, params, x, layer_type, activation, eps_fn):
    """One stochastic forward of a whole model = ModuleWrapper.forward (layers/misc.py:16-25).

    params: dict name -> dict(W_mu, W_rho, bias_mu, bias_rho) (numpy).  eps_fn(name, kind, shape)
    returns the noise for that tensor (kind in 'W', 'bias', 'act') so that any noise source can be
    replayed.  Returns (logits, kl) with kl summed over layers in module order."""
    act = softplus_act if activation == "softplus" else relu_act
    h = np.asarray(x, F32)
    kl = 0.0
    for op in TOPOLOGY[net_type]:
        if op[0] == "act":
            h = act(h)
        elif op[0] == "pool":
            h = maxpool2d(h, op[1], op[2])
        elif op[0] == "flatten":
            h = h.reshape(-1, op[1])
        else:
            p = params[op[1]]
            if layer_type == "bbb":
                We = eps_fn(op[1], "W", p["W_mu"].shape)
                be = eps_fn(op[1], "bias", p["bias_mu"].shape)
                if op[0] == "conv":
                    h, Ws, bs = bbb_conv2d_forward(h, p["W_mu"], p["W_rho"], p["bias_mu"], p["bias_rho"],
                                                   We, be, op[4], op[5], 1)
                else:
                    h, Ws, bs = bbb_linear_forward(h, p["W_mu"], p["W_rho"], p["bias_mu"], p["bias_rho"], We, be)
            else:
                if op[0] == "conv":
                    am, av = lrt_moments_conv2d(h, p["W_mu"], p["W_rho"], p["bias_mu"], p["bias_rho"],
                                                op[4], op[5], 1)
                else:
                    am, av = lrt_moments_linear(h, p["W_mu"], p["W_rho"], p["bias_mu"], p["bias_rho"])
                h = lrt_output(am, av, eps_fn(op[1], "act", am.shape))
                Ws, bs = sigma_from_rho(p["W_rho"]), sigma_from_rho(p["bias_rho"])
            kl += kl_loss(p["W_mu"], Ws, params["_prior_mu"], params["_prior_sigma"])
            kl += kl_loss(p["bias_mu"], bs, params["_prior_mu"], params["_prior_sigma"])
    return h, kl


# ----------------------------------------------------------------------------------------------
# bf16 STORAGE model (BASELINE.json configs[1]).  The reference has no reduced-precision mode: this is the same
# algorithm (layers/BBB/BBBConv.py:61-77, models/BayesianModels/*.py) with the rounding points the HIP bf16 path
# documents in include/bbb_hip.h -- sampled weight matrices, the input image and every hidden activation are rounded
# once to bf16 (nearest-even) where they are stored; biases, accumulation, the epilogue (bias + activation), KL and
# the logits stay fp32.
# ----------------------------------------------------------------------------------------------
def bf16_round(a):
    """fp32 -> nearest-even bf16, returned as fp32 values."""
    u = np.ascontiguousarray(np.asarray(a, F32)).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(F32).reshape(np.shape(a))


def model_forward_bf16(net_type, params, x, activation, eps_fn):
    """model_forward(layer_type='bbb') under the bf16 storage model.  Returns (logits fp32, kl)."""
    act = softplus_act if activation == "softplus" else relu_act
    ops_ = TOPOLOGY[net_type]
    last = max(i for i, op in enumerate(ops_) if op[0] in ("conv", "fc"))
    h = bf16_round(x)
    kl = 0.0
    i = 0
    while i < len(ops_):
        op = ops_[i]
        if op[0] == "act":
            h = bf16_round(act(h))                       # unfused activation (not produced by the topologies above)
        elif op[0] == "pool":
            h = maxpool2d(h, op[1], op[2])
        elif op[0] == "flatten":
            h = h.reshape(-1, op[1])
        else:
            p = params[op[1]]
            Ws, bs = sigma_from_rho(p["W_rho"]), sigma_from_rho(p["bias_rho"])
            w = bf16_round(reparam(p["W_mu"], Ws, eps_fn(op[1], "W", p["W_mu"].shape)))
            b = reparam(p["bias_mu"], bs, eps_fn(op[1], "bias", p["bias_mu"].shape))
            y = conv2d(h, w, b, op[4], op[5], 1) if op[0] == "conv" else linear(h, w, b)
            fused = i + 1 < len(ops_) and ops_[i + 1][0] == "act"
            if fused:                                    # activation is part of the GEMM epilogue, applied in fp32
                y = act(y)
                i += 1
            h = y if i >= last else bf16_round(y)
            kl += kl_loss(p["W_mu"], Ws, params["_prior_mu"], params["_prior_sigma"])
            kl += kl_loss(p["bias_mu"], bs, params["_prior_mu"], params["_prior_sigma"])
        i += 1
    return h, kl
