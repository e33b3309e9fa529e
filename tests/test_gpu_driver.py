"""N4 on hardware: the shape of the reference's train_model / validate_model loops (main_bayesian.py:33-86) driven through the
drop-in `layers` on the MI355X, against numbers recorded from the UNMODIFIED loops run on the reference's own CPU layers
(tests/golden/driver.npz, written by tests/golden/make_golden.py::make_driver in the build container).

The GPU box has no upstream checkout, so main_bayesian.py itself cannot be imported here; the two loops below restate its
control flow line by line (test infrastructure), while everything they call -- `net(inputs)`, `kl_loss`, autograd, the
optimizer step -- is the product.  Noise: the layers' replay hook draws from torch's CPU generator in the reference's order,
so iteration by iteration the same eps is used as in the recording.  Run with -m gpu."""
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


def cpu_eps(shape):
    return torch.empty(tuple(shape)).normal_(0, 1)


def logmeanexp(x, dim):                                   # utils.py:14-22
    x_max, _ = torch.max(x, dim, keepdim=True)
    return (x_max + torch.log(torch.mean(torch.exp(x - x_max), dim, keepdim=True))).squeeze(dim)


def train_model(env, net, optimizer, criterion, loader, num_ens, beta_type, epoch, num_epochs, log):
    """main_bayesian.py:33-62."""
    M = env["metrics"]
    net.train()
    training_loss, accs, kl_list = 0.0, [], []
    for i, (inputs, labels) in enumerate(loader, 1):
        optimizer.zero_grad()
        inputs, labels = inputs.cuda(), labels.cuda()
        outputs = torch.zeros(inputs.shape[0], net.num_classes, num_ens, device="cuda")
        kl = 0.0
        for j in range(num_ens):
            net_out, _kl = net(inputs)
            kl += _kl
            outputs[:, :, j] = F.log_softmax(net_out, dim=1)
        kl = kl / num_ens
        kl_list.append(kl.item())
        log_outputs = logmeanexp(outputs, dim=2)
        beta = M.get_beta(i - 1, len(loader), beta_type, epoch, num_epochs)
        loss = criterion(log_outputs, labels, kl, beta)
        loss.backward()
        optimizer.step()
        accs.append(M.acc(log_outputs.data, labels).item())
        training_loss += loss.item()
        log.append((loss.item(), kl.item(), beta, F.nll_loss(log_outputs.detach().double(), labels).item(), accs[-1]))
    return training_loss / len(loader), np.mean(accs), np.mean(kl_list)


def validate_model(env, net, criterion, loader, num_ens, beta_type, epoch, num_epochs, log):
    """main_bayesian.py:65-86 (note net.train(): sampling stays on)."""
    M = env["metrics"]
    net.train()
    valid_loss, accs = 0.0, []
    for i, (inputs, labels) in enumerate(loader):
        inputs, labels = inputs.cuda(), labels.cuda()
        outputs = torch.zeros(inputs.shape[0], net.num_classes, num_ens, device="cuda")
        kl = 0.0
        for j in range(num_ens):
            net_out, _kl = net(inputs)
            kl += _kl
            outputs[:, :, j] = F.log_softmax(net_out, dim=1).data
        log_outputs = logmeanexp(outputs, dim=2)
        beta = M.get_beta(i - 1, len(loader), beta_type, epoch, num_epochs)
        v = criterion(log_outputs, labels, kl, beta).item()
        valid_loss += v
        accs.append(M.acc(log_outputs, labels).item())
        log.append((v, float(kl), beta, F.nll_loss(log_outputs.detach().double(), labels).item(), accs[-1]))
    return valid_loss / len(loader), np.mean(accs)


@pytest.mark.parametrize("lt", ["bbb", "lrt"])
def test_reference_loops_on_the_dropin_layers(env, golden_driver, lt):
    D = golden_driver
    eps_seed, NB, BS, E = (int(v) for v in D["meta"])
    net = env["zoo"].getModel("lenet", 1, 10, P.CONFIG_PRIORS, lt, "softplus")
    sd = {k[len("init."):]: torch.from_numpy(D[k]) for k in D.files if k.startswith("init.")}
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    for m in net.modules():
        if hasattr(m, "eps_source"):
            m.eps_source = cpu_eps
    loader = [(torch.from_numpy(D[f"{lt}.x"][b]), torch.from_numpy(D[f"{lt}.y"][b])) for b in range(NB)]
    criterion = env["metrics"].ELBO(NB * BS).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    tlog, vlog = [], []
    torch.manual_seed(eps_seed)
    tl, ta, tk = train_model(env, net, opt, criterion, loader, E, "Blundell", 0, 1, tlog)
    vl, va = validate_model(env, net, criterion, loader, E, 0.1, 0, 1, vlog)
    want_t, want_v = D[f"{lt}.train_iter"], D[f"{lt}.valid_iter"]
    got_t, got_v = np.array(tlog), np.array(vlog)
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
    np.testing.assert_allclose([tl, tk], D[f"{lt}.train_ret"][[0, 2]], rtol=2e-6)
    np.testing.assert_allclose(vl, D[f"{lt}.valid_ret"][0], rtol=2e-6)
    assert abs(ta - D[f"{lt}.train_ret"][1]) <= 1.0 / BS and abs(va - D[f"{lt}.valid_ret"][1]) <= 1.0 / BS
    # parameters after three Adam steps: first / second moments of every tensor
    for k, v in net.state_dict().items():
        a = v.detach().double().cpu().numpy().ravel()
        w = D[f"{lt}.final.{k}"]
        np.testing.assert_allclose([np.abs(a).sum(), (a * a).sum()], w[1:3], rtol=1e-4)
        np.testing.assert_allclose(a[:64], w[3:3 + min(64, a.size)], rtol=0, atol=2.5e-3)    # each element moved <= 3 * lr


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
