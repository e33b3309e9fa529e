"""Training extension on the MI355X: multi-tensor Adam vs torch.optim.Adam, and the reference's
train_model batch loop (main_bayesian.py:36-62) -- autograd through the HIP kernels + FusedAdam -- against the CPU port
of the reference (oracle/ref_port_torch.train_steps) with the noise replayed from torch's CPU generator.
Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo, train
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo, train=train)


def test_fused_adam_matches_torch_adam(env):
    """Same parameters, same gradients, 6 steps: parameters and both moments agree with torch.optim.Adam to 2 ulp-ish
    (rtol 2e-6 + a few ulp of the tensor scale where terms cancel); state dicts are interchangeable."""
    torch.manual_seed(0)
    shapes = [(64, 3, 11, 11), (64,), (10, 33), (1,), (1025,), (7, 5, 3)]
    a = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    fa = env["train"].FusedAdam(a, lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
    ta = torch.optim.Adam(b, lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
    for it in range(6):
        for x, y in zip(a, b):
            g = torch.randn_like(x) * (10.0 ** (it - 3))
            x.grad, y.grad = g.clone(), g.clone()
        fa.step()
        ta.step()
    for x, y in zip(a, b):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=1e-8)   # atol: a few ulp of the lr-sized update
        for key in ("exp_avg", "exp_avg_sq"):       # fused multiply-add contraction may differ: 1-2 ulp of the tensor's scale
            want = ta.state[y][key].cpu().numpy()
            np.testing.assert_allclose(fa.state[x][key].cpu().numpy(), want, rtol=2e-6, atol=4e-7 * float(np.abs(want).max()))
    ta2 = torch.optim.Adam(a, lr=3e-3)
    ta2.load_state_dict(fa.state_dict())            # interchangeable state
    assert int(ta2.state[a[0]]["step"]) == 6


def _cpu_eps(shape):
    return torch.empty(tuple(shape)).normal_(0, 1)


@pytest.mark.parametrize("net_type,layer_type", [("lenet", "bbb"), ("lenet", "lrt")])
def test_training_loop_matches_reference_port(env, net_type, layer_type):
    """3 iterations of train_model's loop, num_ens = 2, eps replayed in the reference's draw order.  Step-1 gradients
    (rtol 3e-4 of each tensor's scale), per-step losses (rtol 1e-4) and the parameters after 3 Adam steps (an Adam update
    is +-lr per element, so a gradient whose sign is inside rounding noise may differ by 2*lr: at most 0.5 % of the
    elements may differ by more than 1e-5)."""
    B, E, lr, beta, train_size, ncls = 8, 2, 1e-3, 0.1, 1000.0, 10
    torch.manual_seed(42)
    params = P.init_params(net_type, 1, ncls, P.CONFIG_PRIORS)
    batches = [(torch.rand(B, 1, 32, 32), torch.randint(0, ncls, (B,))) for _ in range(3)]
    net = env["zoo"].getModel(net_type, 1, ncls, P.CONFIG_PRIORS, layer_type, "softplus")
    net.load_state_dict({f"{n}.{k}": v.clone() for n, p in params.items() if not n.startswith("_") for k, v in p.items()})
    net = net.cuda().train()
    for m in net.modules():
        if hasattr(m, "eps_source"):
            m.eps_source = _cpu_eps
    opt = env["train"].FusedAdam(net.parameters(), lr=lr)
    names = [n for n, _ in net.named_parameters()]
    torch.manual_seed(7)
    losses, first_grads = [], None
    for x, y in batches:                                   # the reference's own loop shape (main_bayesian.py:40-58)
        opt.zero_grad()
        outs, kl = [], 0.0
        for j in range(E):
            net_out, _kl = net(x.cuda())
            kl = kl + _kl
            outs.append(F.log_softmax(net_out, dim=1))
        kl = kl / E
        log_outputs = torch.logsumexp(torch.stack(outs, 2), 2) - np.log(E)
        loss = env["train"].elbo(log_outputs, y.cuda(), kl, beta, train_size)
        loss.backward()
        if first_grads is None:
            first_grads = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}
        opt.step()
        losses.append(loss.item())
    # reference side: same seeds, CPU autograd + torch.optim.Adam
    ref = {n: {k: v.clone() for k, v in p.items()} if isinstance(p, dict) else p for n, p in params.items()}
    torch.manual_seed(7)
    probe = {n: {k: v.clone().requires_grad_(True) for k, v in p.items()} if isinstance(p, dict) else p for n, p in params.items()}
    # step-1 gradients of the port (consumes the same eps as the first iteration)
    x0, y0 = batches[0]
    outputs = torch.zeros(B, ncls, E)
    kl = 0.0
    for j in range(E):
        o, k = P.forward(net_type, probe, x0, layer_type, "softplus")
        kl = kl + k
        outputs[:, :, j] = F.log_softmax(o, dim=1)
    (F.nll_loss(P.logmeanexp(outputs, 2), y0, reduction="mean") * train_size + beta * kl / E).backward()
    for n in names:
        lname, k = n.split(".")
        want = probe[lname][k].grad
        got = first_grads[n]
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 3e-4 * scale + 1e-7, (n, float((got - want).abs().max()), scale)
    torch.manual_seed(7)
    ref_losses = P.train_steps(net_type, ref, batches, ncls, E, lr, beta, train_size, layer_type, "softplus")
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-4)
    tot = bad = 0
    for n, p in net.named_parameters():
        lname, k = n.split(".")
        d = (p.detach().cpu() - ref[lname][k].detach()).abs()
        assert float(d.max()) <= 2 * 3 * lr + 1e-6               # never further apart than every step going the other way
        tot += d.numel()
        bad += int((d > 1e-5).sum())
    print("params differing by > 1e-5:", bad, "of", tot)
    assert bad <= 0.005 * tot, (bad, tot)


def test_train_step_helper_learns_and_counts_calls(env):
    """train.train_step (Philox noise, batched draws): loss goes down on a fixed batch; the noise call counter advances
    by num_ens per step."""
    torch.manual_seed(1)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    env["rng"].manual_seed(5, call=0)
    opt = env["train"].FusedAdam(net.parameters(), lr=1e-3)
    x = torch.rand(64, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (64,), device="cuda")
    losses = []
    for it in range(25):
        loss, lo, kl = env["train"].train_step(net, opt, x, y, 2, 1e-7, 64.0)
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    assert env["rng"].get_state()[1] == 2 * 25
    assert lo.shape == (64, 10) and float(lo.exp().sum(1).sub(1).abs().max()) < 1e-3


@pytest.mark.parametrize("layer_type", ["bbb", "lrt"])
def test_graphed_train_step_equals_eager_steps(env, layer_type):
    """GraphedTrainStep (forward + backward + noise counter + Adam in one hipGraph, step count and call counter on the
    device) reproduces the eager train_step sequence: same noise calls, same losses, same parameters."""
    T = env["train"]
    B, E, lr, beta, n = 32, 2, 1e-3, 0.1, 1000.0
    torch.manual_seed(3)
    x = torch.rand(B, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (B,), device="cuda")

    def fresh():
        torch.manual_seed(11)
        net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, layer_type, "softplus").cuda()
        env["rng"].assign_stream_ids(net)
        env["rng"].manual_seed(77, call=0)
        return net

    net_e = fresh()
    opt_e = T.FusedAdam(net_e.parameters(), lr=lr)
    eager_losses = [T.train_step(net_e, opt_e, x, y, E, beta, n)[0].item() for _ in range(6)]
    net_g = fresh()
    opt_g = T.FusedAdam(net_g.parameters(), lr=lr, capturable=True)
    g = T.GraphedTrainStep(net_g, opt_g, x, y, E, beta, n, warmup=3)          # 3 real iterations, then capture
    graph_losses = []
    for _ in range(3):
        loss, lo, kl = g.step()
        graph_losses.append(loss.item())
    np.testing.assert_allclose(graph_losses, eager_losses[3:], rtol=1e-5)
    for (na, a), (nb, b) in zip(net_e.named_parameters(), net_g.named_parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=na)
    assert env["rng"].get_state()[1] == 6 * E
    assert float(opt_g.state[next(iter(net_g.parameters()))]["step"].item()) == 6.0
    # a new batch through the static buffers
    loss2, _, _ = g.step(torch.rand_like(x), y)
    assert np.isfinite(loss2.item())


@pytest.mark.parametrize("layer_type", ["bbb", "lrt"])
def test_train_step_captures_itself_and_keeps_the_eager_sequence(env, layer_type):
    """train_step's default: after auto_graph["after"] identical calls the step runs as one hipGraph.  Same losses and parameters
    as graph=False; a validation pass in between (it draws noise: the host call counter moves) is followed; a new batch shape
    drops the graph."""
    T = env["train"]
    B, E, lr, beta, n = 32, 2, 1e-3, 0.1, 1000.0
    torch.manual_seed(3)
    x = torch.rand(B, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (B,), device="cuda")
    xv = torch.rand(B, 1, 32, 32, device="cuda")

    def run(graph):
        torch.manual_seed(11)
        net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, layer_type, "softplus").cuda()
        env["rng"].assign_stream_ids(net)
        env["rng"].manual_seed(77, call=0)
        opt = T.FusedAdam(net.parameters(), lr=lr)
        losses, captured = [], []
        for it in range(9):
            if it == 6:
                with torch.no_grad():
                    net(xv)                                            # consumes one noise call outside the training step
            losses.append(T.train_step(net, opt, x, y, E, beta * (1 + it), n, graph=graph)[0].item())
            st = T._auto.get(net)
            captured.append(bool(st and st["graphed"] is not None))
        return net, opt, losses, captured

    net_e, opt_e, eager, cap_e = run(False)
    net_g, opt_g, auto, cap_g = run(None)
    assert not any(cap_e)
    assert cap_g == [False] * 3 + [True] * 6                              # calls 1-3 eager, call 4 eager on the capture stream + capture
    np.testing.assert_allclose(auto, eager, rtol=2e-5)
    for (na, a), (nb, b) in zip(net_e.named_parameters(), net_g.named_parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=na)
    assert env["rng"].get_state()[1] == 9 * E + 1
    assert float(opt_g.state[next(iter(net_g.parameters()))]["step"].item()) == 9.0
    # returned tensors are the caller's own (not the graph's static outputs)
    l1 = T.train_step(net_g, opt_g, x, y, E, beta, n)[0]
    v1 = l1.item()
    T.train_step(net_g, opt_g, x, y, E, beta, n)
    assert l1.item() == v1
    # another batch shape: back to launch-by-launch steps, graph dropped
    T.train_step(net_g, opt_g, x[:16], y[:16], E, beta, n)
    assert T._auto[net_g]["graphed"] is None
    # a forward hook is something a replay would skip: never captured
    h = net_g.register_forward_hook(lambda *a: None)
    for _ in range(6):
        T.train_step(net_g, opt_g, x, y, E, beta, n)
    assert T._auto[net_g]["graphed"] is None
    h.remove()


def test_train_step_captures_despite_stale_gradient_accumulators(env):
    """A non-detached output of an earlier forward + backward on the default stream is still referenced: its graph keeps the
    parameters' gradient accumulators (bound to that stream) alive, and a capture on another stream that reached them would make
    autograd synchronise with it (observed: a segmentation fault inside capture_end).  The captured step is rooted at fresh
    leaves sharing the parameters' storage, so the capture goes through and reproduces the launch-by-launch sequence."""
    import torch.nn.functional as F
    T = env["train"]
    x = torch.rand(32, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")
    keep = []

    def run(graph):
        torch.manual_seed(3)
        net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
        env["rng"].assign_stream_ids(net)
        env["rng"].manual_seed(9, call=0)
        lo, kl = env["ens"].mc_forward(net, x, 2, kl_mode="mean")
        (F.nll_loss(lo, y) + 1e-3 * kl).backward()              # lo / kl stay referenced (keep)
        keep.append((lo, kl))
        opt = T.FusedAdam(net.parameters(), lr=1e-3)
        losses = [T.train_step(net, opt, x, y, 2, 0.1, 1000.0, graph=graph)[0].item() for _ in range(8)]
        return net, losses

    net_e, eager = run(False)
    net_g, auto = run(None)
    assert T._auto[net_g]["graphed"] is not None
    np.testing.assert_allclose(auto, eager, rtol=2e-5)
    for a, b in zip(net_e.parameters(), net_g.parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=1e-7)
    assert all(lo.requires_grad for lo, _ in keep)


def test_two_param_groups_capturable_and_self_captured(env):
    """The usual (mu-group, rho-group) split: two groups whose 'params' lists have equal length.  Capturable FusedAdam looks its
    group up by INDEX (list.index compared the dicts by value: `tensor == tensor` raised, or picked the wrong group's lr), so
    the per-group learning rates reach the kernel -- eagerly, with capturable=True, and through train_step's self-capture."""
    T = env["train"]
    x = torch.rand(32, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")

    def run(mode):
        torch.manual_seed(5)
        net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
        env["rng"].assign_stream_ids(net)
        env["rng"].manual_seed(21, call=0)
        mus = [p for n, p in net.named_parameters() if n.endswith("_mu")]
        rhos = [p for n, p in net.named_parameters() if n.endswith("_rho")]
        assert len(mus) == len(rhos)
        opt = T.FusedAdam([{"params": mus, "lr": 1e-3}, {"params": rhos, "lr": 3e-2}], capturable=(mode == "capturable"))
        losses = [T.train_step(net, opt, x, y, 2, 0.1, 1000.0, graph=(None if mode == "auto" else False))[0].item() for _ in range(7)]
        return net, losses

    net_e, eager = run("eager")
    net_c, cap = run("capturable")
    net_a, auto = run("auto")
    assert T._auto[net_a]["graphed"] is not None
    np.testing.assert_allclose(cap, eager, rtol=2e-5)
    np.testing.assert_allclose(auto, eager, rtol=2e-5)
    for (n, a), b, c in zip(net_e.named_parameters(), net_c.parameters(), net_a.parameters()):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=n)
        np.testing.assert_allclose(c.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=n)
    # the rho group really moved 30x faster than the mu group would have
    d_rho = (net_e.conv1.W_rho.detach() - (-5.0)).abs().mean().item()
    assert d_rho > 0.05


def test_loaded_optimizer_state_and_moved_storage_drop_the_capture(env):
    """optimizer.load_state_dict() replaces exp_avg / exp_avg_sq / step with new tensors, `p.data = ...` moves a parameter's
    storage: a step captured before either would keep reading and writing the OLD buffers.  Both are part of train_step's
    capture key: the graph is dropped and later steps follow the loaded state / the new storage."""
    T = env["train"]
    x = torch.rand(32, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")
    torch.manual_seed(5)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    env["rng"].manual_seed(21, call=0)
    opt = T.FusedAdam(net.parameters(), lr=1e-3)
    for _ in range(5):
        T.train_step(net, opt, x, y, 2, 0.1, 1000.0)
    assert T._auto[net]["graphed"] is not None
    import copy
    sd = copy.deepcopy(opt.state_dict())
    for st in sd["state"].values():
        st["exp_avg"].zero_()
        st["exp_avg_sq"].fill_(1.0)
    opt.load_state_dict(sd)
    p0 = net.conv1.W_mu
    before = p0.detach().clone()
    T.train_step(net, opt, x, y, 2, 0.1, 1000.0)
    st = T._auto.get(net)
    assert st is None or st["graphed"] is None                         # key changed: launch by launch again
    # with exp_avg = 0 and exp_avg_sq = 1 loaded, one Adam step moves every element by far less than lr
    assert (p0.detach() - before).abs().max().item() < 1e-3
    assert opt.state[p0]["exp_avg_sq"].mean().item() > 0.9            # the LOADED moments were updated, not the old ones
    for _ in range(5):
        T.train_step(net, opt, x, y, 2, 0.1, 1000.0)
    assert T._auto[net]["graphed"] is not None
    p0.data = p0.data.clone()                                          # new storage for one parameter
    T.train_step(net, opt, x, y, 2, 0.1, 1000.0)
    st = T._auto.get(net)
    assert st is None or st["graphed"] is None


def test_parameter_versions_follow_the_hip_updates(env):
    """bbb_adam_step writes parameters through raw pointers; FusedAdam.step and every replay of a captured training step bump
    Tensor._version, so the version-keyed caches and guards (speculation cache of layers/_fused.py, stale-backward check) see it."""
    T = env["train"]
    x = torch.rand(32, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")
    torch.manual_seed(5)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    opt = T.FusedAdam(net.parameters(), lr=1e-3)
    seen = []
    for _ in range(7):
        T.train_step(net, opt, x, y, 1, 0.1, 1000.0)
        seen.append(net.conv1.W_mu._version)
    assert T._auto[net]["graphed"] is not None
    assert all(b > a for a, b in zip(seen, seen[1:])), seen
    # the "parameter changed before a delayed backward" check can fire now: a forward recorded before an optimizer step must not
    # be differentiated after it (its backward re-reads the live parameters)
    lo, kl = env["ens"].mc_forward(net, x, 2, kl_mode="mean")
    T.train_step(net, opt, x, y, 1, 0.1, 1000.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        (F.nll_loss(lo, y) + 1e-3 * kl).backward()
