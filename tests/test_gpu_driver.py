"""N4 on hardware: the reference's OWN driver on the MI355X over the drop-in `layers`.

The unmodified upstream files reach the GPU box as oracle/_ref/upstream_snapshot.zip (packed byte for byte by
__graft_entry__.build(), oracle/ref_snapshot.py; conftest unpacks it to a temporary directory).  Two things run here:

  * `main_bayesian.train_model` / `validate_model` THEMSELVES (imported through run_reference.prepare, nothing restated) on
    cuda:0, iteration by iteration against numbers recorded from the same functions on the reference's own CPU layers
    (tests/golden/driver.npz, tests/golden/make_golden.py::make_driver).  Noise: the layers' replay hook draws from torch's CPU
    generator in the reference's order, so every iteration uses the eps of the recording.
  * `python main_bayesian.py --net_type N --dataset D` literally (runpy, `__main__`), one epoch on synthetic data, and the
    checkpoint it writes loaded back -- strictly -- into the upstream model built on the UPSTREAM layers.
Run with -m gpu."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import metrics, rng, zoo, train
    return dict(metrics=metrics, rng=rng, zoo=zoo, train=train)


_UPSTREAM_MODULES = ("main_bayesian", "config_bayesian", "metrics", "utils", "data", "models", "uncertainty_estimation",
                     "torchvision")


def _is_upstream(name):
    return any(name == m or name.startswith(m + ".") for m in _UPSTREAM_MODULES)


@pytest.fixture()
def upstream(reference_dir):
    """The interpreter state run_reference.prepare() sets up, undone afterwards (upstream's top-level module names -- utils,
    metrics, data, models -- must not leak into the other test modules of this process)."""
    if reference_dir is None:
        pytest.skip("no upstream files (neither /root/reference nor oracle/_ref/upstream_snapshot.zip)")
    saved_path, saved_mods, saved_argv = list(sys.path), dict(sys.modules), list(sys.argv)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau
    import run_reference as rr
    yield rr, reference_dir
    torch.optim.lr_scheduler.ReduceLROnPlateau = sched
    sys.path[:] = saved_path
    sys.argv[:] = saved_argv
    for k in [k for k in sys.modules if k not in saved_mods and _is_upstream(k)]:
        del sys.modules[k]


def cpu_eps(shape):
    return torch.empty(tuple(shape)).normal_(0, 1)


@pytest.mark.reference
@pytest.mark.parametrize("lt", ["bbb", "lrt"])
def test_reference_loops_on_the_dropin_layers(env, upstream, golden_driver, lt):
    rr, ref = upstream
    mb = rr.prepare(ref, synthetic=64)
    import layers
    assert mb.__file__.startswith(ref) and not layers.__file__.startswith(ref) and str(mb.device) == "cuda:0"
    import metrics as ref_metrics                      # upstream's metrics.py (ELBO, acc, get_beta as the loops call them)
    assert ref_metrics.__file__.startswith(ref)
    D = golden_driver
    eps_seed, NB, BS, E = (int(v) for v in D["meta"])
    net = mb.getModel("lenet", 1, 10, P.CONFIG_PRIORS, lt, "softplus")        # upstream model class over the drop-in layers
    assert type(net).__module__.startswith("models.BayesianModels") and type(net.conv1).__module__.startswith("layers.")
    sd = {k[len("init."):]: torch.from_numpy(D[k]) for k in D.files if k.startswith("init.")}
    net.load_state_dict(sd, strict=True)
    net = net.to(mb.device)
    for m in net.modules():
        if hasattr(m, "eps_source"):
            m.eps_source = cpu_eps
    loader = [(torch.from_numpy(D[f"{lt}.x"][b]), torch.from_numpy(D[f"{lt}.y"][b])) for b in range(NB)]
    elbo = ref_metrics.ELBO(NB * BS).to(mb.device)
    seen = []

    class Recording(torch.nn.Module):                  # the recorder of make_golden.make_driver, verbatim in behaviour
        def forward(self, inp, target, kl, beta):
            v = elbo(inp, target, kl, beta)
            nll = F.nll_loss(inp.detach().double(), target)
            acc = (inp.detach().argmax(1) == target).double().mean()
            seen.append((float(v.detach()), float(kl), float(beta), float(nll), float(acc)))
            return v

    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    torch.manual_seed(eps_seed)
    tl, ta, tk = mb.train_model(net, opt, Recording(), loader, num_ens=E, beta_type="Blundell", epoch=0, num_epochs=1)
    n_train = len(seen)
    vl, va = mb.validate_model(net, Recording(), loader, num_ens=E, beta_type=0.1, epoch=0, num_epochs=1)
    want_t, want_v = D[f"{lt}.train_iter"], D[f"{lt}.valid_iter"]
    got_t, got_v = np.array(seen[:n_train]), np.array(seen[n_train:])
    print(f"[driver {lt}] train nll got {got_t[:, 3]} want {want_t[:, 3]}; valid nll got {got_v[:, 3]} want {want_v[:, 3]}")
    # measured on the MI355X: every iteration's NLL agrees with the recording to ~1e-7 relative, also after the Adam steps
    # (same eps, fp32 everywhere); the bounds below leave a factor ~50
    np.testing.assert_allclose(got_t[:, 1], want_t[:, 1], rtol=2e-6)            # kl (mean over the ensemble)
    np.testing.assert_allclose(got_t[:, 2], want_t[:, 2], rtol=0, atol=0)       # beta schedule (Blundell)
    np.testing.assert_allclose(got_t[:, 3], want_t[:, 3], rtol=5e-6)            # nll, before and after parameter updates
    np.testing.assert_allclose(got_t[:, 0], want_t[:, 0], rtol=2e-6)            # ELBO (dominated by beta * kl)
    np.testing.assert_allclose(got_v[:, 1], want_v[:, 1], rtol=2e-6)            # validation: kl summed over the ensemble
    np.testing.assert_allclose(got_v[:, 3], want_v[:, 3], rtol=5e-6)
    np.testing.assert_allclose(got_v[:, 0], want_v[:, 0], rtol=2e-6)
    assert np.abs(got_t[:, 4] - want_t[:, 4]).max() <= 1.0 / BS + 1e-9          # accuracy: at most one argmax flip per batch
    np.testing.assert_allclose([float(tl), float(tk)], D[f"{lt}.train_ret"][[0, 2]], rtol=2e-6)
    np.testing.assert_allclose(float(vl), D[f"{lt}.valid_ret"][0], rtol=2e-6)
    assert abs(ta - D[f"{lt}.train_ret"][1]) <= 1.0 / BS and abs(va - D[f"{lt}.valid_ret"][1]) <= 1.0 / BS
    # parameters after three Adam steps: first / second moments of every tensor
    for k, v in net.state_dict().items():
        a = v.detach().double().cpu().numpy().ravel()
        w = D[f"{lt}.final.{k}"]
        np.testing.assert_allclose([np.abs(a).sum(), (a * a).sum()], w[1:3], rtol=1e-4)
        np.testing.assert_allclose(a[:64], w[3:3 + min(64, a.size)], rtol=0, atol=2.5e-3)    # each element moved <= 3 * lr


@pytest.mark.reference
@pytest.mark.parametrize("net_type,dataset,lt,bs,n", [("lenet", "MNIST", "bbb", 64, 320), ("alexnet", "CIFAR10", "lrt", 128, 640),
                                                      ("3conv3fc", "CIFAR100", "lrt", 64, 320)])
def test_literal_main_bayesian_runs_and_its_checkpoint_loads_upstream(upstream, tmp_path, monkeypatch, capfd, net_type, dataset, lt, bs, n):
    """`python main_bayesian.py --net_type N --dataset D`, the file itself as __main__ on cuda:0 (main_bayesian.py:89-142):
    one epoch of train_model + validate_model, ReduceLROnPlateau, the 'Validation loss decreased' branch and torch.save.  The
    checkpoint must carry the reference's keys and load strictly into the upstream model on the UPSTREAM layers (CPU)."""
    rr, ref = upstream
    monkeypatch.chdir(tmp_path)                         # checkpoints/<dataset>/bayesian/ is relative to the working directory
    from bbb_hip import _lib
    _lib.lib()
    g = rr.run_literal(ref, net_type, dataset, synthetic=n,
                       overrides={"n_epochs": 1, "layer_type": lt, "batch_size": bs, "train_ens": 2, "valid_ens": 2, "num_workers": 0})
    out = capfd.readouterr().out
    assert "Epoch: 0" in out and "Saving model" in out, out[-2000:]
    import re
    nums = [float(v) for v in re.findall(r"(?:Training Loss|Validation Loss|train_kl_div): ([-+0-9.eEinfa]+)", out)]
    assert len(nums) == 3 and all(np.isfinite(nums)) and nums[2] > 0, out[-2000:]
    assert g["__name__"] == "__main__" and g["__file__"].startswith(ref) and str(g["device"]) == "cuda:0"
    ck = tmp_path / "checkpoints" / dataset / "bayesian" / f"model_{net_type}_{lt}_softplus.pt"
    assert ck.is_file()
    sd = torch.load(str(ck), map_location="cpu")
    assert all(k.rsplit(".", 1)[1] in ("W_mu", "W_rho", "bias_mu", "bias_rho") for k in sd)
    assert all(torch.isfinite(v).all() for v in sd.values())
    # a model of the UPSTREAM layers (not ours) accepts it strictly: swap `layers` for the upstream package in a child module space
    import subprocess
    cin, ncls = {"MNIST": (1, 10), "CIFAR10": (3, 10), "CIFAR100": (3, 100)}[dataset]
    code = r"""
import sys; sys.dont_write_bytecode = True
sys.path.insert(0, %r)
import torch
from unittest import mock
import layers
assert layers.__file__.startswith(%r)
from models.BayesianModels.BayesianLeNet import BBBLeNet
from models.BayesianModels.BayesianAlexNet import BBBAlexNet
from models.BayesianModels.Bayesian3Conv3FC import BBB3Conv3FC
import config_bayesian as cfg
cls = {"lenet": BBBLeNet, "alexnet": BBBAlexNet, "3conv3fc": BBB3Conv3FC}[%r]
with mock.patch("torch.cuda.is_available", return_value=False):
    net = cls(%d, %d, cfg.priors, %r, "softplus")
sd = torch.load(%r, map_location="cpu")
net.load_state_dict(sd, strict=True)
out, kl = net(torch.rand(4, %d, 32, 32))
assert out.shape == (4, %d) and torch.isfinite(out).all() and torch.isfinite(kl)
print("LOADED_OK")
""" % (ref, ref, net_type, ncls, cin, lt, str(ck), cin, ncls)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0 and "LOADED_OK" in r.stdout, r.stderr[-3000:]


def test_device_side_acc_and_beta(env):
    M = env["metrics"]
    out = torch.tensor([[0.1, 0.9], [0.8, 0.2], [0.3, 0.7], [0.6, 0.4]], device="cuda")
    tgt = torch.tensor([1, 0, 0, 0], device="cuda")
    a = M.acc(out, tgt)
    assert a.is_cuda and a.dim() == 0 and abs(a.item() - 0.75) < 1e-7
    meter = M.AccMeter()
    meter.update(out, tgt)
    meter.update(out, 1 - tgt)
    assert abs(meter.mean() - 0.5) < 1e-7
    assert M.get_beta(0, 4, "Blundell") == 2 ** 3 / 15 and M.get_beta(3, 4, "Standard") == 0.25
    assert M.get_beta(1, 4, 0.1) == 0.1 and M.get_beta(0, 4, "Soenderby", 1, 8) == 0.5 and M.get_beta(0, 4, None) == 0
    with pytest.raises(ValueError):
        M.get_beta(0, 4, "Soenderby")


def test_graphed_train_step_follows_lr_and_beta_changes(env):
    """A captured training step must see optimizer.param_groups[...]['lr'] (ReduceLROnPlateau upstream) and a per-batch beta
    (get_beta) change AFTER capture: both live in device scalars the graph reads."""
    T = env["train"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("lenet", 1, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(16, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (16,), device="cuda")
    opt = T.FusedAdam(net.parameters(), lr=1e-3, capturable=True)
    g = T.GraphedTrainStep(net, opt, x, y, num_ens=1, beta=0.1, train_size=16, warmup=2)
    loss_a, _, kl_a = g.step()
    torch.cuda.synchronize()
    la, ka = loss_a.item(), kl_a.item()
    loss_b, _, kl_b = g.step(beta=0.0)                       # ELBO without the KL term: the loss must collapse to the NLL part
    torch.cuda.synchronize()
    assert loss_b.item() < 1e-3 * la and abs(kl_b.item() - ka) < 5e-2 * ka     # KL itself moves ~1 % per Adam step
    p0 = net.conv1.W_mu.detach().clone()
    opt.param_groups[0]["lr"] = 0.0                          # what a scheduler does between epochs
    g.step(beta=0.1)
    torch.cuda.synchronize()
    assert torch.equal(net.conv1.W_mu.detach(), p0)          # lr = 0 -> no movement: the replay read the new rate
    opt.param_groups[0]["lr"] = 1e-2
    g.step()
    torch.cuda.synchronize()
    d = (net.conv1.W_mu.detach() - p0).abs().max().item()
    assert 1e-3 < d <= 1.2e-2                                # Adam moves each weight by about lr
