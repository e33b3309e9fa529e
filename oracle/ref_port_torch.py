"""CPU ORACLE (test infrastructure only) -- functional torch-CPU port of the reference's MC step.

Checker / CPU baseline only: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product path never imports or falls back to it.

Why a second oracle next to ``bbb_numpy.py``: this one executes the *same ATen operators in the
same order* as the upstream nn.Modules (``torch.empty(..).normal_`` from the default CPU
generator, ``log1p(exp(rho))``, ``F.conv2d`` / ``F.linear``, the 10-op KL expression), so that
  (a) under the same ``torch.manual_seed`` it reproduces the unmodified reference bit-for-bit
      (checked in the build container by tests/test_oracle_golden.py::test_port_matches_live_reference
      and against the committed fixtures everywhere), and
  (b) timing it on the GPU box's host cores is a faithful stand-in for "the reference's own CPU
      nn.Module path" (bench.py cpu_baseline, kind="port"), since /root/reference does not exist
      there.
It is written as plain functions over a parameter dict; the model graphs come from the topology
table in ``bbb_numpy.TOPOLOGY``.

Reference lines followed: layers/BBB/BBBConv.py:61-83, layers/BBB/BBBLinear.py:54-76,
layers/BBB_LRT/BBBConv.py:62-87, layers/BBB_LRT/BBBLinear.py:56-79, metrics.py:12-14,27-29,
layers/misc.py:16-25,34-35, utils.py:14-22, main_bayesian.py:43-53,73-80.
"""
import torch
import torch.nn.functional as F

from bbb_numpy import TOPOLOGY

DEFAULT_PRIORS = {  # layers/BBB/BBBConv.py:29-35
    "prior_mu": 0, "prior_sigma": 0.1,
    "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-3, 0.1),
}
CONFIG_PRIORS = {  # config_bayesian.py:4-9
    "prior_mu": 0, "prior_sigma": 0.1,
    "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1),
}


def init_params(net_type, in_channels, n_classes, priors=None):
    """Draw parameters exactly as constructing the upstream model does: per Bayesian layer, in
    definition order, W_mu, W_rho, bias_mu, bias_rho each by ``.normal_(mean, std)`` from the
    default generator (reset_parameters, layers/BBB/BBBConv.py:53-59)."""
    priors = priors or DEFAULT_PRIORS
    mu0, rho0 = priors["posterior_mu_initial"], priors["posterior_rho_initial"]
    params = {}
    cin = in_channels
    feat = None
    for op in TOPOLOGY[net_type]:
        if op[0] == "conv":
            shape = (op[2], cin, op[3], op[3])
            cin = op[2]
        elif op[0] == "flatten":
            feat = op[1]
            continue
        elif op[0] == "fc":
            out = op[2] if op[2] is not None else n_classes
            shape = (out, feat)
            feat = out
        else:
            continue
        p = {}
        p["W_mu"] = torch.empty(shape).normal_(*mu0)
        p["W_rho"] = torch.empty(shape).normal_(*rho0)
        p["bias_mu"] = torch.empty(shape[0]).normal_(*mu0)
        p["bias_rho"] = torch.empty(shape[0]).normal_(*rho0)
        params[op[1]] = p
    params["_prior_mu"] = priors["prior_mu"]
    params["_prior_sigma"] = priors["prior_sigma"]
    return params


def _kl(mu_q, sig_q, mu_p, sig_p):
    # metrics.py:28, verbatim operator order (arguments arrive swapped from the call sites)
    return 0.5 * (2 * torch.log(sig_p / sig_q) - 1 + (sig_q / sig_p).pow(2) + ((mu_p - mu_q) / sig_p).pow(2)).sum()


def _bbb_layer(h, p, conv_args, sample, eps=None):
    """eps: None = the reference's own source (torch's default CPU generator, W first then bias); or a callable
    (kind, shape) -> tensor that replays another noise stream (the device's Philox stream in the parity tests)."""
    if sample:
        W_eps = torch.empty(p["W_mu"].size()).normal_(0, 1) if eps is None else eps("W", tuple(p["W_mu"].shape))
        W_sigma = torch.log1p(torch.exp(p["W_rho"]))
        weight = p["W_mu"] + W_eps * W_sigma
        bias_eps = torch.empty(p["bias_mu"].size()).normal_(0, 1) if eps is None else eps("bias", tuple(p["bias_mu"].shape))
        bias_sigma = torch.log1p(torch.exp(p["bias_rho"]))
        bias = p["bias_mu"] + bias_eps * bias_sigma
    else:
        W_sigma = torch.log1p(torch.exp(p["W_rho"]))
        bias_sigma = torch.log1p(torch.exp(p["bias_rho"]))
        weight, bias = p["W_mu"], p["bias_mu"]
    if conv_args is not None:
        y = F.conv2d(h, weight, bias, conv_args[0], conv_args[1], 1, 1)
    else:
        y = F.linear(h, weight, bias)
    return y, W_sigma, bias_sigma


def _lrt_layer(h, p, conv_args, sample, eps=None):
    W_sigma = torch.log1p(torch.exp(p["W_rho"]))
    bias_sigma = torch.log1p(torch.exp(p["bias_rho"]))
    bias_var = bias_sigma ** 2
    if conv_args is not None:
        act_mu = F.conv2d(h, p["W_mu"], p["bias_mu"], conv_args[0], conv_args[1], 1, 1)
        act_var = 1e-16 + F.conv2d(h ** 2, W_sigma ** 2, bias_var, conv_args[0], conv_args[1], 1, 1)
    else:
        act_mu = F.linear(h, p["W_mu"], p["bias_mu"])
        act_var = 1e-16 + F.linear(h ** 2, W_sigma ** 2, bias_var)
    act_std = torch.sqrt(act_var)
    if sample:
        e = torch.empty(act_mu.size()).normal_(0, 1) if eps is None else eps("act", tuple(act_mu.shape))
        return act_mu + act_std * e, W_sigma, bias_sigma
    return act_mu, W_sigma, bias_sigma


def forward(net_type, params, x, layer_type="bbb", activation="softplus", sample=True, eps_fn=None):
    """One stochastic forward of the whole model -> (logits, kl).  eps_fn(layer_name, kind, shape) -> tensor replays an
    external noise stream (kind in 'W', 'bias', 'act'); None = torch's CPU generator, as upstream."""
    act = F.softplus if activation == "softplus" else F.relu
    layer = _bbb_layer if layer_type == "bbb" else _lrt_layer
    h = x
    sig = []
    for op in TOPOLOGY[net_type]:
        if op[0] == "act":
            h = act(h)
        elif op[0] == "pool":
            h = F.max_pool2d(h, op[1], op[2])
        elif op[0] == "flatten":
            h = h.view(-1, op[1])
        else:
            p = params[op[1]]
            eps = None if eps_fn is None else (lambda kind, shape, _n=op[1]: eps_fn(_n, kind, shape))
            h, Ws, bs = layer(h, p, (op[4], op[5]) if op[0] == "conv" else None, sample, eps)
            sig.append((p, Ws, bs))
    kl = 0.0
    for p, Ws, bs in sig:  # layers/misc.py:20-23 + kl_loss of each layer
        k = _kl(params["_prior_mu"], params["_prior_sigma"], p["W_mu"], Ws)
        k += _kl(params["_prior_mu"], params["_prior_sigma"], p["bias_mu"], bs)
        kl = kl + k
    return h, kl


def logmeanexp(x, dim):
    x_max, _ = torch.max(x, dim, keepdim=True)  # utils.py:20-21
    return (x_max + torch.log(torch.mean(torch.exp(x - x_max), dim, keepdim=True))).squeeze(dim)


def mc_step(net_type, params, x, n_classes, num_ens, layer_type="bbb", activation="softplus"):
    """main_bayesian.py:73-80: E x net(x) -> log_softmax -> outputs[:, :, j]; logmeanexp(dim=2).
    Returns (log_outputs [B, C], kl summed over the E calls as validate_model does)."""
    outputs = torch.zeros(x.shape[0], n_classes, num_ens)
    kl = 0.0
    for j in range(num_ens):
        net_out, _kl_j = forward(net_type, params, x, layer_type, activation)
        kl += _kl_j
        outputs[:, :, j] = F.log_softmax(net_out, dim=1)
    return logmeanexp(outputs, 2), kl


def train_steps(net_type, params, batches, n_classes, num_ens, lr, beta, train_size, layer_type="bbb", activation="softplus"):
    """The batch loop of train_model (main_bayesian.py:36-62) on the functional port, CPU autograd + torch.optim.Adam
    (main_bayesian.py:103): for each (x, y): zero_grad; num_ens forwards; kl / num_ens; logmeanexp; ELBO
    (metrics.py:19-24: nll_loss(mean) * train_size + beta * kl); backward; step.  Mutates `params` in place
    (leaf tensors get requires_grad).  Returns the list of losses."""
    leaves = []
    for name, p in params.items():
        if name.startswith("_"):
            continue
        for k in ("W_mu", "W_rho", "bias_mu", "bias_rho"):
            p[k].requires_grad_(True)
            leaves.append(p[k])
    opt = torch.optim.Adam(leaves, lr=lr)
    losses = []
    for x, y in batches:
        opt.zero_grad()
        outputs = torch.zeros(x.shape[0], n_classes, num_ens)
        kl = 0.0
        for j in range(num_ens):
            net_out, _kl_j = forward(net_type, params, x, layer_type, activation)
            kl = kl + _kl_j
            outputs[:, :, j] = F.log_softmax(net_out, dim=1)
        kl = kl / num_ens
        log_outputs = logmeanexp(outputs, 2)
        loss = F.nll_loss(log_outputs, y, reduction="mean") * train_size + beta * kl
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses
