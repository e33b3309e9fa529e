#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Run in the build container only (it needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference has no golden vectors of its own, so parity is pinned on
outputs of the reference modules themselves, executed on CPU in fp32 with fixed torch seeds:
  layers_small.npz  -- the four layer types on tiny shapes: params, input, replayed eps, output, KL
  functions.npz     -- metrics.calculate_kl / ELBO / get_beta, utils.logmeanexp on fixed inputs
  driver.npz        -- the UNMODIFIED main_bayesian.train_model / validate_model (main_bayesian.py:33-86) run on a synthetic
                       loader: initial state_dict, the batches, per-iteration ELBO values and the functions' return values
  models.npz        -- whole-model forwards (LeNet with full params; AlexNet / 3Conv3FC by seed +
                       parameter checksums + expected logits / KL), one MC-ensemble step, and the
                       3x224x224 AlexNet case whose logits come out as [B*49, classes].
Replay protocol for eps: save the CPU generator state before the forward, run it,
restore the state and redraw torch.empty(shape).normal_(0, 1) in the layer's order.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn.functional as F

import layers as ref_layers          # noqa: E402  (the reference's package)
import metrics as ref_metrics        # noqa: E402
import utils as ref_utils            # noqa: E402
import config_bayesian as ref_cfg    # noqa: E402
from models.BayesianModels.BayesianLeNet import BBBLeNet          # noqa: E402
from models.BayesianModels.BayesianAlexNet import BBBAlexNet      # noqa: E402
from models.BayesianModels.Bayesian3Conv3FC import BBB3Conv3FC    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def npy(t):
    return t.detach().cpu().numpy().copy()


def replay(draw_shapes, state):
    cur = torch.get_rng_state()
    torch.set_rng_state(state)
    out = [torch.empty(s).normal_(0, 1) for s in draw_shapes]
    torch.set_rng_state(cur)
    return out


def layer_case(out, tag, layer, x, kind):
    """kind: 'bbb' (draws W then bias) or 'lrt' (one activation-shaped draw)."""
    layer.train()
    state = torch.get_rng_state()
    y = layer(x)
    kl = layer.kl_loss()
    if kind == "bbb":
        shapes = [layer.W_mu.shape] + ([layer.bias_mu.shape] if layer.use_bias else [])
    else:
        shapes = [y.shape]
    eps = replay(shapes, state)
    out[f"{tag}.x"] = npy(x)
    out[f"{tag}.W_mu"] = npy(layer.W_mu)
    out[f"{tag}.W_rho"] = npy(layer.W_rho)
    if layer.use_bias:
        out[f"{tag}.bias_mu"] = npy(layer.bias_mu)
        out[f"{tag}.bias_rho"] = npy(layer.bias_rho)
    out[f"{tag}.eps0"] = npy(eps[0])
    if len(eps) > 1:
        out[f"{tag}.eps1"] = npy(eps[1])
    out[f"{tag}.y"] = npy(y)
    out[f"{tag}.kl"] = npy(kl)
    out[f"{tag}.W_sigma"] = npy(layer.W_sigma)
    layer.eval()
    out[f"{tag}.y_nosample"] = npy(layer(x, sample=False))
    layer.train()
    # gradients of (sum(y * g) + 0.37 * kl) with the same eps, for the backward parity tests
    torch.set_rng_state(state)
    xg = x.clone().requires_grad_(True)
    y2 = layer(xg)
    g = torch.linspace(-1, 1, y2.numel()).reshape(y2.shape)
    loss = (y2 * g).sum() + 0.37 * layer.kl_loss()
    layer.zero_grad()
    loss.backward()
    out[f"{tag}.g"] = npy(g)
    out[f"{tag}.grad_x"] = npy(xg.grad)
    out[f"{tag}.grad_W_mu"] = npy(layer.W_mu.grad)
    out[f"{tag}.grad_W_rho"] = npy(layer.W_rho.grad)
    if layer.use_bias:
        out[f"{tag}.grad_bias_mu"] = npy(layer.bias_mu.grad)
        out[f"{tag}.grad_bias_rho"] = npy(layer.bias_rho.grad)


def make_layers():
    out = {}
    torch.manual_seed(101)
    layer_case(out, "bbb_conv", ref_layers.BBB_Conv2d(3, 5, 3, stride=2, padding=1), torch.randn(2, 3, 9, 9), "bbb")
    out["bbb_conv.meta"] = np.array([3, 5, 3, 3, 2, 1, 1, 1])  # cin cout kh kw stride pad dil bias
    torch.manual_seed(102)
    layer_case(out, "bbb_conv_nb", ref_layers.BBB_Conv2d(2, 4, (2, 3), stride=1, padding=2, dilation=2, bias=False),
               torch.randn(3, 2, 8, 7), "bbb")
    out["bbb_conv_nb.meta"] = np.array([2, 4, 2, 3, 1, 2, 2, 0])
    torch.manual_seed(103)
    layer_case(out, "bbb_lin", ref_layers.BBB_Linear(7, 4), torch.randn(5, 7), "bbb")
    torch.manual_seed(104)
    layer_case(out, "lrt_conv", ref_layers.BBB_LRT_Conv2d(4, 6, 3, stride=1, padding=1, priors=ref_cfg.priors),
               torch.rand(3, 4, 6, 6), "lrt")
    out["lrt_conv.meta"] = np.array([4, 6, 3, 3, 1, 1, 1, 1])
    torch.manual_seed(105)
    layer_case(out, "lrt_lin_nb", ref_layers.BBB_LRT_Linear(9, 5, bias=False), torch.randn(4, 9), "lrt")
    torch.manual_seed(106)
    layer_case(out, "lrt_lin", ref_layers.BBB_LRT_Linear(33, 10, priors=ref_cfg.priors), torch.rand(6, 33), "lrt")
    np.savez_compressed(os.path.join(HERE, "layers_small.npz"), **out)
    return out


def make_functions():
    out = {}
    torch.manual_seed(7)
    mu = torch.randn(20000) * 0.1
    rho = torch.randn(20000) * 0.1 - 5
    sig = torch.log1p(torch.exp(rho))
    out["kl.mu"], out["kl.rho"], out["kl.sigma"] = npy(mu), npy(rho), npy(sig)
    out["kl.value_cfg"] = npy(ref_metrics.calculate_kl(0, 0.1, mu, sig))        # call-site order
    out["kl.value_textbook"] = npy(ref_metrics.calculate_kl(mu, sig, torch.tensor(0.0), torch.tensor(0.1)))
    rho3 = torch.randn(20000) * 0.1 - 3
    out["kl.rho3"] = npy(rho3)
    out["kl.value_default"] = npy(ref_metrics.calculate_kl(0, 0.1, mu, torch.log1p(torch.exp(rho3))))
    x = torch.randn(6, 10, 5) * 3
    out["lme.x"] = npy(x)
    out["lme.dim2"] = npy(ref_utils.logmeanexp(x, dim=2))
    out["lme.dim0_keep"] = npy(ref_utils.logmeanexp(x, dim=0, keepdim=True))
    out["lme.none"] = npy(ref_utils.logmeanexp(x))
    lo = F.log_softmax(torch.randn(8, 10), dim=1)
    tgt = torch.randint(0, 10, (8,))
    out["elbo.log_outputs"], out["elbo.target"] = npy(lo), npy(tgt)
    out["elbo.value"] = npy(ref_metrics.ELBO(50000)(lo, tgt, torch.tensor(1234.5), 0.1))
    betas = []
    for bt in ["Blundell", "Soenderby", "Standard", "nonsense", 0.25]:
        betas.append([float(ref_metrics.get_beta(b, 10, bt, 3, 40)) for b in range(4)])
    out["beta.table"] = np.array(betas)
    acc_out = torch.randn(16, 10)
    acc_t = torch.randint(0, 10, (16,))
    out["acc.outputs"], out["acc.targets"] = npy(acc_out), npy(acc_t)
    out["acc.value"] = np.array(ref_metrics.acc(acc_out, acc_t))
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **out)


def checksums(net):
    cs = []
    for _, p in net.named_parameters():
        d = p.detach().double()
        cs.append([d.sum().item(), (d * d).sum().item()])
    return np.array(cs)


def model_case(out, tag, cls, n_classes, in_ch, layer_type, B, hw, seed, priors, full_params, act="softplus"):
    torch.manual_seed(seed)
    net = cls(n_classes, in_ch, priors, layer_type, act)
    x = torch.rand(B, in_ch, hw, hw)
    torch.manual_seed(seed + 1)
    net.train()
    logits, kl = net(x)
    out[f"{tag}.meta"] = np.array([n_classes, in_ch, B, hw, seed])
    out[f"{tag}.logits"] = npy(logits)
    out[f"{tag}.kl"] = npy(kl)
    out[f"{tag}.checksums"] = checksums(net)
    out[f"{tag}.x_checksum"] = np.array(x.double().sum().item())
    if full_params:
        out[f"{tag}.x"] = npy(x)
        for k, v in net.state_dict().items():
            out[f"{tag}.sd.{k}"] = npy(v)
    return net, x


def make_models():
    out = {}
    P = ref_cfg.priors
    model_case(out, "lenet_bbb", BBBLeNet, 10, 1, "bbb", 4, 32, 11, P, True)
    model_case(out, "lenet_lrt", BBBLeNet, 10, 1, "lrt", 4, 32, 12, None, True, act="relu")
    model_case(out, "alexnet_bbb", BBBAlexNet, 10, 3, "bbb", 3, 32, 13, P, False)
    model_case(out, "alexnet_lrt", BBBAlexNet, 100, 3, "lrt", 3, 32, 14, P, False)
    model_case(out, "3conv3fc_bbb", BBB3Conv3FC, 10, 3, "bbb", 2, 32, 15, P, False)
    model_case(out, "3conv3fc_lrt", BBB3Conv3FC, 10, 3, "lrt", 2, 32, 16, P, False, act="relu")
    model_case(out, "alexnet224_bbb", BBBAlexNet, 10, 3, "bbb", 1, 224, 17, P, False)

    # one MC-ensemble step exactly as validate_model / train_model do it (main_bayesian.py:43-53,73-80)
    torch.manual_seed(21)
    net = BBBLeNet(10, 1, P, "bbb", "softplus")
    x = torch.rand(4, 1, 32, 32)
    labels = torch.randint(0, 10, (4,))
    torch.manual_seed(22)
    E = 3
    outputs = torch.zeros(4, 10, E)
    kl = 0.0
    per_draw = []
    for j in range(E):
        net_out, _kl = net(x)
        kl += _kl
        per_draw.append(npy(net_out))
        outputs[:, :, j] = F.log_softmax(net_out, dim=1).data
    log_outputs = ref_utils.logmeanexp(outputs, dim=2)
    out["mc_lenet.meta"] = np.array([10, 1, 4, 32, 21, E])
    out["mc_lenet.logits"] = np.stack(per_draw)
    out["mc_lenet.log_outputs"] = npy(log_outputs)
    out["mc_lenet.kl_sum"] = npy(kl)
    out["mc_lenet.labels"] = npy(labels)
    out["mc_lenet.elbo_valid"] = npy(ref_metrics.ELBO(1000)(log_outputs, labels, kl, 0.1))
    out["mc_lenet.elbo_train"] = npy(ref_metrics.ELBO(1000)(log_outputs, labels, kl / E, 0.1))
    out["mc_lenet.checksums"] = checksums(net)
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)


def make_uncertainty():
    """uncertainty_estimation.get_uncertainty_per_image on CPU (its module imports data/torchvision/seaborn at the top,
    so the function is executed from its source text with those imports stubbed; the function body is untouched)."""
    import types
    src = open(os.path.join(REF, "uncertainty_estimation.py")).read()
    start = src.index("def get_uncertainty_per_image")
    end = src.index("def get_uncertainty_per_batch")
    ns = {"torch": torch, "np": np, "F": F}
    exec(compile(src[start:end], "uncertainty_estimation.py", "exec"), ns)
    fn = ns["get_uncertainty_per_image"]
    out = {}
    for tag, lt, seed in (("lrt", "lrt", 31), ("bbb", "bbb", 32)):
        torch.manual_seed(seed)
        net = BBBLeNet(10, 1, ref_cfg.priors, lt, "softplus")
        img = torch.rand(1, 32, 32)
        T = 15
        for norm in (False, True):
            torch.manual_seed(seed + 1)
            logits, _ = net(img.unsqueeze(0).repeat(T, 1, 1, 1))
            torch.manual_seed(seed + 1)
            pred, epi, ale = fn(net, img, T=T, normalized=norm)
            k = f"unc_{tag}_{int(norm)}"
            out[k + ".logits"] = npy(logits)
            out[k + ".pred"], out[k + ".epistemic"], out[k + ".aleatoric"] = pred, epi, ale
    np.savez_compressed(os.path.join(HERE, "uncertainty.npz"), **out)


def make_driver():
    """Per-iteration numbers of the reference's own training / validation loop (not a restatement of it): main_bayesian is
    imported unmodified (torchvision stubbed: it is not installed and the loop does not use it), its train_model and
    validate_model run on CPU with the reference's layers.  The noise is the default CPU generator's stream after
    torch.manual_seed(EPS_SEED): a consumer that draws torch.empty(shape).normal_(0, 1) in the layers' order (W, then bias,
    layer by layer, forward by forward) replays it exactly."""
    import types
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    import main_bayesian as mb
    out = {}
    EPS_SEED, NB, BS, E = 4321, 3, 8, 2
    for lt in ("bbb", "lrt"):
        torch.manual_seed(77)
        net = mb.getModel("lenet", 1, 10, ref_cfg.priors, lt, "softplus")
        for k, v in net.state_dict().items():      # same seed, same shapes: one copy serves both layer types
            if lt == "bbb":
                out[f"init.{k}"] = npy(v)
            else:
                assert np.array_equal(out[f"init.{k}"], npy(v))
        g = torch.Generator().manual_seed(5)
        batches = [(torch.rand(BS, 1, 32, 32, generator=g), torch.randint(0, 10, (BS,), generator=g)) for _ in range(NB)]
        out[f"{lt}.x"] = np.stack([npy(b[0]) for b in batches])
        out[f"{lt}.y"] = np.stack([npy(b[1]) for b in batches])
        elbo = ref_metrics.ELBO(NB * BS)
        seen = []

        class Recording(torch.nn.Module):           # records what the unmodified loop passes to / gets from its criterion
            def forward(self, inp, target, kl, beta):
                v = elbo(inp, target, kl, beta)
                nll = F.nll_loss(inp.detach().double(), target)       # the part of the ELBO that depends on the outputs
                acc = (inp.detach().argmax(1) == target).double().mean()
                seen.append((float(v.detach()), float(kl), float(beta), float(nll), float(acc)))
                return v

        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        torch.manual_seed(EPS_SEED)
        tl, ta, tk = mb.train_model(net, opt, Recording(), batches, num_ens=E, beta_type="Blundell", epoch=0, num_epochs=1)
        n_train = len(seen)
        vl, va = mb.validate_model(net, Recording(), batches, num_ens=E, beta_type=0.1, epoch=0, num_epochs=1)
        out[f"{lt}.train_iter"] = np.array(seen[:n_train], dtype=np.float64)        # [NB, (loss, kl, beta, nll, acc)]
        out[f"{lt}.valid_iter"] = np.array(seen[n_train:], dtype=np.float64)
        out[f"{lt}.train_ret"] = np.array([float(tl), float(ta), float(tk)], dtype=np.float64)
        out[f"{lt}.valid_ret"] = np.array([float(vl), float(va)], dtype=np.float64)
        for k, v in net.state_dict().items():      # parameters after the 3 Adam steps: moments + a 64-element window
            a = npy(v).astype(np.float64).ravel()
            out[f"{lt}.final.{k}"] = np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[:64]])
    out["meta"] = np.array([EPS_SEED, NB, BS, E], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "driver.npz"), **out)


if __name__ == "__main__":
    if "--driver-only" in sys.argv:
        make_driver()
        print("driver.npz", os.path.getsize(os.path.join(HERE, "driver.npz")), "bytes")
        sys.exit(0)
    make_driver()
    make_uncertainty()
    make_layers()
    make_functions()
    make_models()
    for f in ("layers_small.npz", "functions.npz", "models.npz", "uncertainty.npz", "driver.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
