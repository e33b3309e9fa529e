"""Whole-layer / whole-model parity on the MI355X: the drop-in `layers` package and the batched MC-ensemble
path against fixtures generated from the unmodified reference (eps replayed from torch's CPU generator,
SURVEY.md section 4.2) and against the oracle.  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bbb_numpy as O
import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(layers=layers, ops=ops, rng=rng, ens=ensemble, zoo=zoo)


def cpu_eps(shape):
    return torch.empty(tuple(shape)).normal_(0, 1)


def load_params(net, params):
    sd = {}
    for name, p in params.items():
        if name.startswith("_"):
            continue
        for k, v in p.items():
            sd[f"{name}.{k}"] = v
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def set_eps_source(net, fn):
    for m in net.modules():
        if hasattr(m, "eps_source"):
            m.eps_source = fn


# ---------------------------------------------------------------- single layers vs reference fixtures
def _mk_layer(env, L, tag, cls, *args, **kw):
    layer = cls(*args, **kw)
    sd = {k: torch.from_numpy(L[f"{tag}.{k}"]) for k in ("W_mu", "W_rho", "bias_mu", "bias_rho") if f"{tag}.{k}" in L.files}
    layer.load_state_dict(sd)
    return layer.cuda()


LAYER_CASES = [
    ("bbb_conv", "BBB_Conv2d", (3, 5, 3), dict(stride=2, padding=1), None),
    ("bbb_conv_nb", "BBB_Conv2d", (2, 4, (2, 3)), dict(stride=1, padding=2, dilation=2, bias=False), None),
    ("bbb_lin", "BBB_Linear", (7, 4), {}, None),
    ("lrt_conv", "BBB_LRT_Conv2d", (4, 6, 3), dict(stride=1, padding=1), "cfg"),
    ("lrt_lin_nb", "BBB_LRT_Linear", (9, 5), dict(bias=False), None),
    ("lrt_lin", "BBB_LRT_Linear", (33, 10), {}, "cfg"),
]


@pytest.mark.parametrize("tag,cls,args,kw,pri", LAYER_CASES)
def test_layer_forward_kl_and_grads_vs_reference(env, golden, tag, cls, args, kw, pri):
    L = golden["layers_small"]
    if pri:
        kw = dict(kw, priors=P.CONFIG_PRIORS)
    layer = _mk_layer(env, L, tag, getattr(env["layers"], cls), *args, **kw)
    eps = [L[f"{tag}.eps0"]] + ([L[f"{tag}.eps1"]] if f"{tag}.eps1" in L.files else [])
    it = iter(eps)
    layer.eps_source = lambda shape: torch.from_numpy(next(it)).reshape(shape)
    x = torch.from_numpy(L[f"{tag}.x"]).cuda().requires_grad_(True)
    layer.train()
    y = layer(x)
    kl = layer.kl_loss()
    np.testing.assert_allclose(y.detach().cpu().numpy(), L[f"{tag}.y"], rtol=2e-5, atol=2e-6)
    assert abs(kl.item() - float(L[f"{tag}.kl"])) <= 2e-6 * abs(float(L[f"{tag}.kl"]))
    np.testing.assert_allclose(layer.W_sigma.detach().cpu().numpy(), L[f"{tag}.W_sigma"], rtol=1e-6)
    # backward of sum(y*g) + 0.37*kl, same eps (HIP reparam backward + declared ATen stop-gap for dgrad/wgrad)
    g = torch.from_numpy(L[f"{tag}.g"]).cuda()
    ((y * g).sum() + 0.37 * kl).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), L[f"{tag}.grad_x"], rtol=2e-4, atol=2e-5)
    for k in ("W_mu", "W_rho", "bias_mu", "bias_rho"):
        if f"{tag}.grad_{k}" in L.files:
            want = L[f"{tag}.grad_{k}"]
            got = getattr(layer, k).grad.cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=3e-4, atol=3e-4 * max(1.0, np.abs(want).max() * 1e-2))
    # deterministic path: eval() + sample=False
    layer.eval()
    layer.eps_source = None
    y0 = layer(x.detach(), sample=False)
    np.testing.assert_allclose(y0.detach().cpu().numpy(), L[f"{tag}.y_nosample"], rtol=2e-5, atol=2e-6)
    assert abs(layer.kl_loss().item() - float(L[f"{tag}.kl"])) <= 2e-6 * abs(float(L[f"{tag}.kl"]))


def test_eval_does_not_turn_sampling_off(env):
    """Reference quirk (layers/BBB/BBBConv.py:61-62): forward(x) samples even in eval()."""
    layer = env["layers"].BBB_Linear(16, 8).cuda().eval()
    x = torch.randn(4, 16, device="cuda")
    a, b = layer(x), layer(x)
    assert not torch.equal(a, b)
    assert torch.equal(layer(x, sample=False), layer(x, sample=False))


# ---------------------------------------------------------------- whole models vs reference fixtures
MODEL_CASES = [("lenet_bbb", "lenet", "bbb", "softplus", True), ("lenet_lrt", "lenet", "lrt", "relu", False),
               ("alexnet_bbb", "alexnet", "bbb", "softplus", True), ("alexnet_lrt", "alexnet", "lrt", "softplus", True),
               ("3conv3fc_bbb", "3conv3fc", "bbb", "softplus", True), ("3conv3fc_lrt", "3conv3fc", "lrt", "relu", True),
               ("alexnet224_bbb", "alexnet", "bbb", "softplus", True)]


@pytest.mark.parametrize("tag,net_type,lt,act,cfgpri", MODEL_CASES)
def test_model_forward_vs_reference(env, golden, tag, net_type, lt, act, cfgpri):
    """Same seed -> same parameters and input as the reference (checked through checksums), eps replayed from
    the CPU generator in the reference's draw order -> logits and KL of the unmodified reference."""
    M = golden["models"]
    ncls, cin, B, hw, seed = [int(v) for v in M[f"{tag}.meta"]]
    pri = P.CONFIG_PRIORS if cfgpri else None
    torch.manual_seed(seed)
    params = P.init_params(net_type, cin, ncls, pri)
    x = torch.rand(B, cin, hw, hw)
    assert abs(x.double().sum().item() - float(M[f"{tag}.x_checksum"])) < 1e-9
    net = load_params(env["zoo"].getModel(net_type, cin, ncls, pri, lt, act), params)
    net.train()
    torch.manual_seed(seed + 1)
    set_eps_source(net, cpu_eps)
    with torch.no_grad():
        logits, kl = net(x.cuda())
    assert logits.shape == M[f"{tag}.logits"].shape
    # fp32 MFMA (sequential fmaf chain over K <= 3456) vs mkldnn's blocked accumulation through up to 8 layers:
    # stated tolerance 5e-4 relative + 2e-5 of the logit scale absolute
    want = M[f"{tag}.logits"]
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(logits.cpu().numpy(), want, rtol=5e-4, atol=2e-5 * scale)
    assert abs(kl.item() - float(M[f"{tag}.kl"])) <= 2e-6 * float(M[f"{tag}.kl"])


def test_mc_step_vs_reference(env, golden):
    """main_bayesian.py:73-83 on LeNet, E=3: log_outputs, summed KL and both ELBO conventions."""
    M = golden["models"]
    torch.manual_seed(21)
    params = P.init_params("lenet", 1, 10, P.CONFIG_PRIORS)
    x = torch.rand(4, 1, 32, 32)
    labels = torch.randint(0, 10, (4,))
    net = load_params(env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus"), params)
    torch.manual_seed(22)
    set_eps_source(net, cpu_eps)
    E = 3
    outs, kl = [], 0.0
    with torch.no_grad():
        for j in range(E):
            o, k = net(x.cuda())
            kl = kl + k
            outs.append(o)
    logits = torch.stack(outs)
    np.testing.assert_allclose(logits.cpu().numpy(), M["mc_lenet.logits"], rtol=5e-4, atol=5e-5)
    lo = env["ops"].mc_tail(logits, mean_over=E)
    np.testing.assert_allclose(lo.cpu().numpy(), M["mc_lenet.log_outputs"], rtol=5e-4, atol=5e-5)
    assert abs(kl.item() - float(M["mc_lenet.kl_sum"])) <= 2e-6 * float(M["mc_lenet.kl_sum"])
    nll = F.nll_loss(lo, labels.cuda(), reduction="mean")
    elbo_valid = nll * 1000 + 0.1 * kl
    elbo_train = nll * 1000 + 0.1 * kl / E
    assert abs(elbo_valid.item() - float(M["mc_lenet.elbo_valid"])) <= 1e-4 * abs(float(M["mc_lenet.elbo_valid"]))
    assert abs(elbo_train.item() - float(M["mc_lenet.elbo_train"])) <= 1e-4 * abs(float(M["mc_lenet.elbo_train"]))


# ---------------------------------------------------------------- batched ensemble == python loop (Philox path)
@pytest.mark.parametrize("net_type,lt,B,hw,cin", [("lenet", "bbb", 8, 32, 1), ("alexnet", "bbb", 16, 32, 3),
                                                   ("3conv3fc", "bbb", 4, 32, 3), ("alexnet", "lrt", 8, 32, 3),
                                                   ("lenet", "lrt", 4, 32, 1)])
def test_batched_ensemble_equals_loop(env, net_type, lt, B, hw, cin):
    torch.manual_seed(3)
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, hw, hw, device="cuda")
    E = 5
    from layers.misc import reference_layout
    with torch.no_grad():
        with reference_layout():                        # the drop-in forward on the reference-layout (NCHW) kernels
            env["rng"].manual_seed(1234, call=10)
            loop = torch.stack([net(x)[0] for _ in range(E)])
            kl_loop = net(x)[1]
        batched, kl = env["ens"].mc_logits(net, x, E, 1234, 10, fuse_act=False)
        fused, _ = env["ens"].mc_logits(net, x, E, 1234, 10, fuse_act=True)
        env["rng"].manual_seed(1234, call=10)           # the drop-in inference forward proper: batch-innermost when B % 4 == 0
        fast_loop = torch.stack([net(x)[0] for _ in range(E)])
    assert torch.equal(loop, batched)                   # same kernels, same k order, same noise calls
    assert torch.equal(fast_loop, fused)                # net(x) under no_grad == draw j of the batched fast path, bitwise
    # batch-innermost path: hardware exp2/log2 softplus in the GEMM epilogue vs torch's softplus between layers
    scale = max(1.0, float(loop.abs().max()))
    np.testing.assert_allclose(fused.cpu().numpy(), loop.cpu().numpy(), rtol=5e-4, atol=2e-5 * scale)
    assert abs(kl.item() - kl_loop.item()) <= 1e-6 * kl_loop.item()
    # the whole step
    env["rng"].manual_seed(1234, call=10)
    with torch.no_grad():
        lo, klsum = env["ens"].mc_forward(net, x, E, fuse_act=False)
    want = O.mc_log_outputs(loop.cpu().numpy())
    np.testing.assert_allclose(lo.cpu().numpy(), want, rtol=2e-5, atol=2e-6)
    assert abs(klsum.item() - E * kl_loop.item()) <= 2e-6 * E * kl_loop.item()
    assert env["rng"].get_state()[1] == 10 + E


def test_simulated_rank_sharding_matches_single_device(env):
    """Draw-sharding without a second GPU: compute each rank's block on this device and combine as
    combine_ranks does; must equal the unsharded step."""
    torch.manual_seed(5)
    net = env["zoo"].BBBAlexNet(10, 3, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(32, 3, 32, 32, device="cuda")
    E, world = 10, 4
    with torch.no_grad():
        full, _ = env["ens"].mc_logits(net, x, E, 77, 0)
        want = env["ops"].mc_tail(full, mean_over=E)
        blocks = []
        for r in range(world):
            lo, hi = env["ens"].draw_range(E, r, world)
            lg, _ = env["ens"].mc_logits(net, x, hi - lo, 77, lo)
            assert torch.equal(lg, full[lo:hi])
            blocks.append(env["ops"].mc_tail(lg, mean_over=0))
        got = torch.logsumexp(torch.stack(blocks), 0) - np.log(E)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-6)


# ---------------------------------------------------------------- statistics of the Philox path
def test_bbb_outputs_follow_lrt_moments(env):
    """For fixed x and parameters the BBB layer's output over draws is N(act_mu, act_var) elementwise, where
    those are exactly the LRT moments -- a known-answer test for RNG + reparam + GEMM together."""
    torch.manual_seed(0)
    pri = dict(P.DEFAULT_PRIORS)
    layer = env["layers"].BBB_Conv2d(3, 8, 3, padding=1, priors=pri).cuda()
    env["rng"].assign_stream_ids(layer)       # (the noise streams of a stand-alone layer otherwise depend on how many layers the process built before)
    x = torch.rand(2, 3, 6, 6, device="cuda")
    E = 4000
    env["rng"].manual_seed(9)
    with torch.no_grad():
        seed, call0 = env["rng"].next_calls(E)
        mus, rhos, ids = layer._param_lists()
        ws, _, _ = env["ops"].reparam_kl_forward(mus, rhos, 0, 0.1, ids, seed, call0, draws=E)
        y = env["ops"].conv2d_forward(x.unsqueeze(0), ws[0], ws[1], 1, 1, 1)          # [E, 2, 8, 6, 6]
    am, av = O.lrt_moments_conv2d(x.cpu().numpy(), *[p.detach().cpu().numpy() for p in
                                                     (layer.W_mu, layer.W_rho, layer.bias_mu, layer.bias_rho)], 1, 1, 1)
    mean = y.mean(0).cpu().numpy()
    var = y.var(0).cpu().numpy()
    z = np.abs(mean - am) / np.sqrt(av / E)
    assert z.max() < 5.0 and np.mean(z) < 1.2            # |mean - act_mu| within sampling error
    ratio = var / av
    # (every output sums the SAME E weight draws: the ratios are correlated, their mean has up to sqrt(2 / E) = 0.022 of spread)
    assert 0.85 < ratio.min() and ratio.max() < 1.15 and abs(ratio.mean() - 1) < 0.04


def test_train_step_runs_and_learns(env):
    """train_model's inner step (main_bayesian.py:40-58) on the batched path: loss decreases."""
    torch.manual_seed(1)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = torch.rand(64, 1, 32, 32, device="cuda")
    y = torch.randint(0, 10, (64,), device="cuda")
    losses = []
    for it in range(30):
        opt.zero_grad()
        lo, kl = env["ens"].mc_forward(net, x, 2, kl_mode="mean")
        loss = F.nll_loss(lo, y, reduction="mean") * 64 + 1e-7 * kl
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_graphed_step_replays_draw_fresh_noise_and_match_eager(env):
    """hipGraph replay r of GraphedMC == eager mc_forward at noise calls call0 + r*E (device-side call counter)."""
    torch.manual_seed(2)
    net = env["zoo"].BBBAlexNet(10, 3, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(64, 3, 32, 32, device="cuda")
    E = 4
    env["rng"].manual_seed(77, call=100)
    g = env["ens"].GraphedMC(net, x, E, streams=2)
    outs = []
    for r in range(3):
        lo, kl = g.step()
        outs.append((lo.clone(), kl.clone()))
    assert env["rng"].get_state() == (77, 100 + 3 * E)
    assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[1][0], outs[2][0])
    with torch.no_grad():
        for r in range(3):
            env["rng"].manual_seed(77, call=100 + r * E)
            lo, kl = env["ens"].mc_forward(net, x, E)
            assert torch.equal(lo, outs[r][0]) and kl.item() == outs[r][1].item()


def test_lrt_graphed_step(env):
    torch.manual_seed(2)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "lrt", "relu").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(16, 1, 32, 32, device="cuda")
    env["rng"].manual_seed(5, call=0)
    g = env["ens"].GraphedMC(net, x, 3)
    a = g.step()[0].clone()
    b = g.step()[0].clone()
    assert not torch.equal(a, b)
    with torch.no_grad():
        env["rng"].manual_seed(5, call=3)
        lo, _ = env["ens"].mc_forward(net, x, 3)
    assert torch.equal(lo, b)


def test_graphed_pipeline_steps_match_eager_sequence(env):
    """Three steps in flight on three streams: step i of the round-robin pipeline == the i-th eager mc_forward."""
    torch.manual_seed(4)
    net = env["zoo"].BBBAlexNet(10, 3, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(64, 3, 32, 32, device="cuda")
    E, depth, nsteps = 4, 3, 7
    env["rng"].manual_seed(31, call=8)
    pipe = env["ens"].GraphedPipeline(net, x, E, depth=depth)
    got = []
    for i in range(nsteps):
        lo, kl = pipe.step()
        pipe.lanes[i % depth].stream.synchronize()
        got.append((lo.clone(), kl.clone()))
    pipe.sync()
    assert env["rng"].get_state() == (31, 8 + nsteps * E)
    with torch.no_grad():
        env["rng"].manual_seed(31, call=8)
        for i in range(nsteps):
            lo, kl = env["ens"].mc_forward(net, x, E)
            assert torch.equal(lo, got[i][0]) and kl.item() == got[i][1].item(), i


def test_graphed_step_multirank_logic_simulated(env, monkeypatch):
    """World-size-2 graphed steps simulated on one GPU: each "rank" captures its own work, packs (lse block, kl share) into its
    send buffer, and its post-graph reduces what the collective delivered.  The collective itself is replaced by a recorder
    that (a) keeps every rank's send buffer and (b) hands the rank a gathered buffer assembled from the ranks simulated so far;
    with both ranks' buffers in hand the reduction must equal the single-device step at the same noise calls, replay after
    replay -- and the post-graph of the LAST simulated rank, which sees real data from both, must return exactly that."""
    import torch.distributed as dist
    ens = env["ens"]
    torch.manual_seed(6)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(32, 1, 32, 32, device="cuda")
    E, world = 5, 2
    sends = {r: [] for r in range(world)}
    cur = {}

    def fake_gather(recv, send, group):
        r, i = cur["rank"], cur["step"]
        sends[r].append(send.clone())
        n = send.numel()
        for q in range(world):                       # ranks already simulated contribute their recorded buffer of step i
            blk = sends[q][i] if len(sends[q]) > i else torch.full_like(send, -float("inf"))
            if len(sends[q]) <= i:
                blk[-1] = 0
            recv[q * n:(q + 1) * n].copy_(blk)

    monkeypatch.setattr(ens, "_all_gather", fake_gather)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: world)
    last_out = []
    for rank in range(world):
        monkeypatch.setattr(dist, "get_rank", lambda group=None, r=rank: r)
        env["rng"].manual_seed(11, call=20)
        g = ens.GraphedMC(net, x, E, group=f"rank{rank}")
        assert (g.lo, g.hi) == ens.draw_range(E, rank, world)
        cur["rank"] = rank
        for i in range(3):
            cur["step"] = i
            lo, kl = g.step()
            torch.cuda.synchronize()
            if rank == world - 1:
                last_out.append((lo.clone(), kl.clone()))
    monkeypatch.undo()
    n = sends[0][0].numel()
    with torch.no_grad():
        for i in range(3):
            env["rng"].manual_seed(11, call=20 + i * E)
            want, kl = ens.mc_forward(net, x, E)
            blocks = torch.stack([sends[q][i][:-1].view(32, 10) for q in range(world)])
            got = torch.logsumexp(blocks, 0) - float(np.log(E))
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=3e-6, atol=3e-6)
            assert abs((sends[0][i][-1] + sends[1][i][-1]).item() - kl.item()) <= 2e-6 * kl.item()
            assert torch.equal(last_out[i][0], got) and abs(last_out[i][1].item() - kl.item()) <= 2e-6 * kl.item()


# ---------------------------------------------------------------- uncertainty estimation (N2)
def test_uncertainty_kernel_vs_reference_fixture(env, golden):
    U = golden["uncertainty"]
    for tag in ("lrt", "bbb"):
        for norm in (0, 1):
            k = f"unc_{tag}_{norm}"
            logits = torch.from_numpy(U[k + ".logits"]).cuda().unsqueeze(1)          # [T, 1, C]
            pred, epi, ale = env["ops"].uncertainty(logits, normalized=bool(norm))
            np.testing.assert_allclose(pred[0].cpu().numpy(), U[k + ".pred"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(epi[0].cpu().numpy(), U[k + ".epistemic"], rtol=1e-3, atol=1e-9)
            np.testing.assert_allclose(ale[0].cpu().numpy(), U[k + ".aleatoric"], rtol=1e-4, atol=1e-8)


def test_uncertainty_per_batch_and_per_image(env):
    from bbb_hip import uncertainty as unc
    torch.manual_seed(8)
    net = env["zoo"].BBBLeNet(10, 1, P.DEFAULT_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    batch = torch.rand(12, 1, 32, 32)
    T = 9
    env["rng"].manual_seed(3, call=0)
    pred, epi, ale = unc.get_uncertainty_per_batch(net, batch, T=T, normalized=False)
    with torch.no_grad():
        logits, _ = env["ens"].mc_logits(net, batch.cuda(), T, 3, 0)
    wp, we, wa = O.uncertainty(logits.cpu().numpy(), False)
    np.testing.assert_allclose(pred, wp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(epi, we, rtol=1e-3, atol=1e-9)
    np.testing.assert_allclose(ale, wa, rtol=1e-4, atol=1e-8)
    assert pred.shape == (12, 10) and (epi >= 0).all() and (ale >= -1e-7).all() and epi.max() > 1e-8
    # per image: T copies in ONE batch -> one weight draw for BBB layers -> no epistemic spread; LRT decorrelates rows
    p1, e1, a1 = unc.get_uncertainty_per_image(net, batch[0].cuda(), T=T)
    assert p1.shape == (10,) and e1.max() < 1e-10
    lrt = env["zoo"].BBBLeNet(10, 1, P.DEFAULT_PRIORS, "lrt", "softplus").cuda()
    p2, e2, a2 = unc.get_uncertainty_per_image(lrt, batch[0].cuda(), T=T, normalized=True)
    assert e2.max() > 1e-9 and np.isfinite(a2).all() and (a2 > -1e-7).all()
    # softmax identity: epistemic + aleatoric = p_bar - p_bar^2 per class, so the class sums stay below 1
    assert 0 < (e1 + a1).sum() < 1 and 0 < (e2 + a2).sum() < 1


def test_pipeline_lanes_own_their_input_buffers(env):
    """GraphedPipeline with a NEW batch every step: each lane copies the batch into its own buffer, so three steps in
    flight never read a batch that a later step is overwriting; results equal the eager per-batch steps bit for bit."""
    torch.manual_seed(8)
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    batches = [torch.rand(64, 1, 32, 32, device="cuda") for _ in range(7)]
    E = 4
    with torch.no_grad():
        env["rng"].manual_seed(99, call=0)
        want = [env["ens"].mc_forward(net, b, E)[0].clone() for b in batches]
        env["rng"].manual_seed(99, call=0)
        pipe = env["ens"].GraphedPipeline(net, batches[0], E, depth=3)
        outs = []
        for b in batches:
            lo, _ = pipe.step(b)
            outs.append(lo)                      # lane buffer: read it before the lane is replayed again (3 steps later)
            if len(outs) >= 3:
                pipe.sync()
        pipe.sync()
        torch.cuda.synchronize()
    # each lane's output buffer holds its LAST step: compare the last three batches, and all of them through a re-run
    for k in range(1, 4):
        assert torch.equal(outs[-k], want[-k])
    env["rng"].manual_seed(99, call=0)
    with torch.no_grad():
        pipe2 = env["ens"].GraphedPipeline(net, batches[0], E, depth=3)
        got = []
        for b in batches:
            lo, _ = pipe2.step(b)
            pipe2.sync()
            got.append(lo.clone())
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_sigma_attributes_come_from_the_fused_pass_and_follow_rho():
    """W_sigma / bias_sigma (readable after forward upstream: layers/BBB/BBBConv.py:64,69): without autograd the fused parameter
    pass's own sigma output, computed on every read -- writes through `.data` (reset_parameters, p.data.copy_) never bump the version
    counter a cache could be keyed on --; with autograd on a trainable rho the differentiable expression."""
    import copy
    import layers
    from bbb_hip import ops
    torch.manual_seed(0)
    l = layers.BBB_Conv2d(3, 8, 3, priors=None).cuda()
    with torch.no_grad():
        s1 = l.W_sigma
        assert torch.equal(s1, l.W_sigma) and not s1.requires_grad
        _, sig, _ = ops.reparam_kl_forward([l.W_mu.detach()], [l.W_rho.detach()], 0, 0.1, [0], 0, 0, draws=1, sample=False,
                                           want_sigma=True, want_kl=False)
        assert torch.equal(s1, sig[0])
        np.testing.assert_allclose(s1.cpu().numpy(), torch.log1p(torch.exp(l.W_rho)).cpu().numpy(), rtol=3e-7)
        l.W_rho.add_(0.5)                                                     # an optimizer step
        s2 = l.W_sigma
        assert s2 is not s1 and float((s2 - s1).abs().min()) > 0
        l.W_rho.data.fill_(-3.0)                                              # a write the version counter does not see
        assert float((l.W_sigma - 0.048587).abs().max()) < 1e-6
        l.reset_parameters()
        np.testing.assert_allclose(l.W_sigma.cpu().numpy(), torch.log1p(torch.exp(l.W_rho)).cpu().numpy(), rtol=3e-7)
        assert l.bias_sigma.shape == (8,) and layers.BBB_Linear(4, 2, bias=False).cuda().bias_sigma is None
    s3 = l.W_sigma                                                            # autograd on: connected to rho
    assert s3.requires_grad
    s3.sum().backward()
    assert l.W_rho.grad is not None and torch.isfinite(l.W_rho.grad).all()
    l2 = copy.deepcopy(l)
    assert "_w_sigma_view" not in l2.__dict__ and torch.equal(l2.W_rho, l.W_rho)


@pytest.mark.parametrize("net_type,lt,shape", [("alexnet", "bbb", (64, 3, 32, 32)), ("alexnet", "lrt", (32, 3, 32, 32)),
                                               ("3conv3fc", "bbb", (16, 3, 32, 32)), ("lenet", "lrt", (8, 1, 32, 32))])
def test_hooked_forward_keeps_the_layout_between_layers(net_type, lt, shape):
    """Forward hooks on children (layers/_fused.hooked_chain): the per-layer path's own kernels without the layout round trip around
    every layer -- logits, KL and every tensor a hook sees are bit for bit those of the module-by-module loop; hooks that REPLACE
    an input (pre-hook) or an output are honoured; hooks on a conv, an activation, a pooling module, the flatten and the classifier."""
    import layers
    from layers import _fused
    from bbb_hip import rng, zoo
    import torch.nn as nn
    torch.manual_seed(5)
    net = zoo.getModel(net_type, shape[1], 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    rng.assign_stream_ids(net)
    x = torch.rand(*shape, device="cuda")
    kids = list(net.children())
    convs = [m for m in kids if hasattr(m, "kernel_size") and hasattr(m, "W_mu")]
    acts = [m for m in kids if isinstance(m, nn.Softplus)]
    pools = [m for m in kids if isinstance(m, nn.MaxPool2d)]
    flat = [m for m in kids if isinstance(m, layers.FlattenLayer)][0]
    last = kids[-1]

    def run(enabled):
        seen = []
        hs = [convs[1].register_forward_hook(lambda m, i, o: seen.append(("conv", i[0].clone(), o.clone()))),
              acts[0].register_forward_hook(lambda m, i, o: seen.append(("act", i[0].clone(), o.clone()))),
              pools[-1].register_forward_hook(lambda m, i, o: o * 0.5),                      # replaces the output
              flat.register_forward_hook(lambda m, i, o: seen.append(("flat", i[0].clone(), o.clone()))),
              last.register_forward_pre_hook(lambda m, i: (i[0] + 1.0,)),                    # replaces the input
              last.register_forward_hook(lambda m, i, o: seen.append(("fc", i[0].clone(), o.clone())))]
        _fused.hooked_chain_enabled[0] = enabled
        try:
            with torch.no_grad():
                rng.manual_seed(11, call=3)
                out, kl = net(x)
                out2, _ = net(x)                                                             # (a second call: the next call index)
        finally:
            _fused.hooked_chain_enabled[0] = True
            rng.use_device_generator()                                                       # (un-pin: later tests seed torch's generator)
            for h in hs:
                h.remove()
        return out, kl, out2, seen

    a, kla, a2, sa = run(True)
    b, klb, b2, sb = run(False)
    assert torch.equal(a, b) and torch.equal(kla, klb) and torch.equal(a2, b2) and not torch.equal(a, a2)
    assert len(sa) == len(sb) == 8
    for (ta, ia, oa), (tb, ib, ob) in zip(sa, sb):
        assert ta == tb and ia.shape == ib.shape and oa.shape == ob.shape and torch.equal(ia, ib) and torch.equal(oa, ob), ta
