"""Training on the batch-innermost kernels (bbb_hip/fast_train.py): gradients against the reference-layout autograd path and
against torch autograd in float64, the pooling / activation backward kernel, and a 3-iteration BayesianAlexNet training run
against the CPU port of the reference's loop (main_bayesian.py:36-62) fed with the device's own Philox noise.  Run with -m gpu."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bbb_numpy as O
import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo, train, fast_train
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo, train=train, ft=fast_train)


@pytest.mark.parametrize("k,s,act,H", [(2, 2, "softplus", 8), (3, 2, "relu", 9), (3, 2, "softplus", 7), (0, 1, "relu", 5),
                                       (2, 2, None, 6), (0, 1, "softplus", 4)])
def test_pool_act_backward_kernel_vs_torch(env, k, s, act, H):
    g = torch.Generator(device="cuda").manual_seed(k * 10 + H)
    planes, B = 6, 12
    v = torch.randn(planes, H, H, B, device="cuda", generator=g) * 2
    if act == "relu":
        v[:, :2] = -v[:, :2].abs()                          # whole rows of exact zeros after ReLU: ties inside windows
    vt = v.permute(3, 0, 1, 2).contiguous().double().requires_grad_(True)      # [B, planes, H, W] for torch
    a = F.softplus(vt) if act == "softplus" else (F.relu(vt) if act == "relu" else vt)
    out = F.max_pool2d(a, k, s) if k else a
    go = torch.randn(out.shape, device="cuda", generator=g, dtype=torch.float64)
    out.backward(go)
    y = (F.softplus(v) if act == "softplus" else (F.relu(v) if act == "relu" else v)).contiguous()
    g_out = go.float().permute(1, 2, 3, 0).contiguous()
    got = env["ops"].pool_act_backward_chwn(g_out, y, k, s, act)
    got_p = env["ops"].pool_act_backward_chwn(g_out, y, k, s, act, pad_planes=True)
    assert torch.equal(got, got_p)
    want = vt.grad.permute(1, 2, 3, 0).float()
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("k,s,act,H,shared", [(2, 2, "softplus", 8, False), (3, 2, "relu", 9, False), (0, 1, None, 4, False),
                                              (0, 1, "softplus", 4, True), (2, 2, "relu", 6, True)])
def test_lrt_pool_act_backward_kernel_vs_torch_float64(env, k, s, act, H, shared):
    """bbb_lrt_pool_act_bwd_chwn: out = pool(act(act_mu + sqrt(act_var) * eps)); gradients w.r.t. act_mu and act_var from torch
    autograd in float64 (layers/BBB_LRT/BBBConv.py:71-81); `shared`: E draws sampled from ONE pair of moments (first layer)."""
    g = torch.Generator(device="cuda").manual_seed(100 + k * 10 + H)
    E, C, B = 3, 4, 8
    Em = 1 if shared else E
    am = torch.randn(Em, C, H, H, B, device="cuda", generator=g)
    av = torch.rand(Em, C, H, H, B, device="cuda", generator=g) * 0.5 + 0.05
    eps = torch.randn(E, C, H, H, B, device="cuda", generator=g)
    am64, av64 = am.double().requires_grad_(True), av.double().requires_grad_(True)
    pre = am64 + torch.sqrt(av64) * eps.double()
    a = F.softplus(pre) if act == "softplus" else (F.relu(pre) if act == "relu" else pre)
    a4 = a.permute(0, 4, 1, 2, 3).reshape(E * B, C, H, H)
    out = F.max_pool2d(a4, k, s) if k else a4
    go = torch.randn(out.shape, device="cuda", generator=g, dtype=torch.float64)
    out.backward(go)
    y = a.detach().float().contiguous()                                        # what the forward kernel stored
    Hp = out.shape[-1]
    g_out = go.float().reshape(E, B, C, Hp, Hp).permute(0, 2, 3, 4, 1).contiguous()
    g_mu, g_var = env["ops"].lrt_pool_act_backward_chwn(g_out, y, am, av, k, s, act)
    g_mu_p, g_var_p = env["ops"].lrt_pool_act_backward_chwn(g_out, y, am, av, k, s, act, pad_planes=True)
    assert torch.equal(g_mu, g_mu_p) and torch.equal(g_var, g_var_p)
    assert torch.equal(g_mu, env["ops"].pool_act_backward_chwn(g_out, y, k, s, act) if (k or act) else g_out)
    if shared:
        g_mu, g_var = g_mu.sum(0, keepdim=True), g_var.sum(0, keepdim=True)
    # the kernel recovers the pre-activation from the fp32 activated output: where softplus saturates towards 0 the
    # recovered value loses relative accuracy exactly as the torch expression the backward used before (y + log(-expm1(-y)))
    np.testing.assert_allclose(g_mu.cpu().numpy(), am64.grad.float().cpu().numpy(), rtol=3e-5, atol=3e-6)
    np.testing.assert_allclose(g_var.cpu().numpy(), av64.grad.float().cpu().numpy(), rtol=2e-3, atol=2e-4)
    v = torch.where(y > 20.0, y, y + torch.log(-torch.expm1(-y))) if act == "softplus" else y
    t = torch.where(y > 0, v - am, torch.zeros_like(v)) if act is not None else v - am
    g_pre = env["ops"].pool_act_backward_chwn(g_out, y, k, s, act) if (k or act) else g_out
    old = g_pre * t / (2.0 * av)                                               # the ATen expression this kernel replaced
    if shared:
        old = old.sum(0, keepdim=True)
    np.testing.assert_allclose(g_var.cpu().numpy(), old.cpu().numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("Cin,Cout,kk,pad,H", [(8, 12, 3, 1, 6), (64, 32, 5, 2, 4), (16, 10, 1, 0, 1)])
def test_chwn_wgrad_dgrad_vs_torch_float64(env, Cin, Cout, kk, pad, H):
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(Cin)
    E, B = 2, 8
    x = torch.randn(E, Cin, H, H, B, device="cuda", generator=g)
    w = torch.randn(E, Cout, Cin, kk, kk, device="cuda", generator=g) * 0.2
    gy = torch.randn(E, Cout, H + 2 * pad - kk + 1, H + 2 * pad - kk + 1, B, device="cuda", generator=g)
    gw = ops.conv2d_chwn_weight_grad(gy, x, tuple(w.shape), 1, pad, 1)
    gx = ops.conv2d_chwn_input_grad(gy, w, (H, H), pad, 1)
    for e in range(E):
        xt = x[e].permute(3, 0, 1, 2).double().requires_grad_(True)
        wt = w[e].double().requires_grad_(True)
        y = F.conv2d(xt, wt, None, 1, pad)
        y.backward(gy[e].permute(3, 0, 1, 2).double())
        np.testing.assert_allclose(gw[e].cpu().numpy(), wt.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(gx[e].cpu().numpy(), xt.grad.permute(1, 2, 3, 0).float().cpu().numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("E,Cin,Cout,kk,pad,H,B,act", [(2, 8, 12, 3, 1, 6, 8, None), (3, 64, 70, 5, 2, 4, 4, "softplus"), (1, 16, 10, 1, 0, 1, 16, None),
                                                         (2, 20, 33, 3, 0, 7, 12, "relu")])
def test_tap_major_weight_operand(env, E, Cin, Cout, kk, pad, H, B, act):
    """bbb_conv_desc_t::w_tap_major: weights given as [E, Cout, kh, kw, Cin], contraction (tap, channel) instead of (channel, tap):
    the same products, so the plain launch's result up to fp32 summation order -- plain and split-contraction launches."""
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(Cin + Cout)
    x = torch.randn(E, Cin, H, H, B, device="cuda", generator=g)
    w = torch.randn(E, Cout, Cin, kk, kk, device="cuda", generator=g) * 0.2
    b = torch.randn(E, Cout, device="cuda", generator=g)
    w_tm = w.permute(0, 1, 3, 4, 2).contiguous()
    for split in (False, True):
        with ops.use_config(split_k=split):
            want = ops.conv2d_chwn_forward(x, w, b, 1, pad, 1, act=act)
            got = ops.conv2d_chwn_forward(x, w_tm, b, 1, pad, 1, act=act, w_tap_major=True)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5)
    with pytest.raises(Exception):
        ops.conv2d_chwn_forward(x, w_tm, b, 1, pad, 1, w_tap_major=True, bf16x3=True)


@pytest.mark.parametrize("E,shared", [(2, False), (3, True)])
def test_chwn_wgrad_reads_the_output_gradient_in_place(env, E, shared):
    """A launch that is not cut into batch chunks CAN read g_pre [E, Cout, Ho, Wo, B] in place as a tap-major weight operand
    (ops.wgrad_in_place, an option): same gradient as the transposed-copy form up to summation order, and as torch's in float64."""
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(40 + E)
    Cin, Cout, kk, pad, H, B = 64, 192, 5, 2, 4, 4                       # (B = 4: no chunks)
    x = torch.randn(1 if shared else E, Cin, H, H, B, device="cuda", generator=g)
    gy = torch.randn(E, Cout, H, H, B, device="cuda", generator=g)
    old = ops.conv2d_chwn_weight_grad(gy, x, (E, Cout, Cin, kk, kk), 1, pad, 1)
    ops.wgrad_in_place[0] = True                                          # (off by default: measured slower on the metric shape)
    try:
        got = ops.conv2d_chwn_weight_grad(gy, x, (E, Cout, Cin, kk, kk), 1, pad, 1)
    finally:
        ops.wgrad_in_place[0] = False
    np.testing.assert_allclose(got.cpu().numpy(), old.cpu().numpy(), rtol=2e-5, atol=2e-5)
    for e in range(E):
        xt = x[0 if shared else e].permute(3, 0, 1, 2).double()
        wt = torch.zeros(Cout, Cin, kk, kk, device="cuda", dtype=torch.float64, requires_grad=True)
        F.conv2d(xt, wt, None, 1, pad).backward(gy[e].permute(3, 0, 1, 2).double())
        np.testing.assert_allclose(got[e].cpu().numpy(), wt.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("E,shared", [(1, False), (3, False), (3, True)])
def test_chwn_wgrad_batch_chunks_vs_torch_float64(env, E, shared):
    """A launch of few workgroups splits the batch into S chunks that run as extra draws: the output gradient lands in the
    chunked weight-operand layout in one pass and the chunk sums fold into the final tap transpose (bbb_transpose_sum_batched)."""
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(E)
    Cin, Cout, kk, pad, H, B = 16, 24, 3, 1, 5, 64
    x = torch.randn(1 if shared else E, Cin, H, H, B, device="cuda", generator=g)
    gy = torch.randn(E, Cout, H, H, B, device="cuda", generator=g)
    S = 4
    gr = ops.chwn_grad_as_weights(gy, S)
    want = gy.permute(0, 1, 4, 2, 3).reshape(E, Cout, S, B // S, H, H).permute(0, 2, 1, 3, 4, 5).reshape(E * S, Cout, B // S, H, H)
    assert torch.equal(gr, want)
    gw = ops.conv2d_chwn_weight_grad(gy, x, (E, Cout, Cin, kk, kk), 1, pad, 1)
    for e in range(E):
        xt = x[0 if shared else e].permute(3, 0, 1, 2).double()
        wt = torch.zeros(Cout, Cin, kk, kk, device="cuda", dtype=torch.float64, requires_grad=True)
        F.conv2d(xt, wt, None, 1, pad).backward(gy[e].permute(3, 0, 1, 2).double())
        np.testing.assert_allclose(gw[e].cpu().numpy(), wt.grad.float().cpu().numpy(), rtol=2e-4, atol=3e-4)


@pytest.mark.parametrize("R,C", [(1, 10), (4, 512), (9, 64), (16, 100), (25, 300), (32, 257), (33, 70), (4, 18001), (9, 8000), (16, 4400),
                                 (32, 2300)])
def test_transposes_with_few_rows(env, R, C):
    """Large operands (>= 2^20 elements) of <= 32 rows landing contiguously, nothing summed, take the few-rows kernel (column reads
    -> LDS -> one linear write), anything else the 32 x 32 tiles: both entries, with batch strides and a summed dimension."""
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(R * 1000 + C)
    a = torch.randn(3, 2, 5, R, C, device="cuda", generator=g)             # [i1][s][i2][r][c]
    out = torch.full((3, 5, C, R), float("nan"), device="cuda")
    assert ops._transpose_sum_batched(a, out, R, C, (3, 5, 1), (2 * 5 * R * C, R * C, 0), (5 * C * R, C * R, 0), C, R, 2, 5 * R * C)
    assert torch.equal(out, (a[:, 0] + a[:, 1]).transpose(2, 3).contiguous())
    b = a[:, 0].contiguous()                                               # [3][5][R][C]
    out2 = torch.full((3, 5, C, R), float("nan"), device="cuda")
    assert ops._transpose_batched(b, out2, R, C, 3, 5, 5 * R * C, R * C, C, 5 * C * R, C * R, R)
    assert torch.equal(out2, b.transpose(2, 3).contiguous())


def test_transpose_sum_batched_entry(env):
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(2, 5, 3, 37, 41, device="cuda", generator=g)          # [i1][s][i2][r][c]
    out = torch.empty(2, 3, 41, 37, device="cuda")
    assert ops._transpose_sum_batched(a, out, 37, 41, (2, 3, 1), (5 * 3 * 37 * 41, 37 * 41, 0), (3 * 41 * 37, 41 * 37, 0), 41, 37,
                                      5, 3 * 37 * 41)
    want = a[:, 0]
    for s_ in range(1, 5):
        want = want + a[:, s_]                                            # ascending order, one rounding per addition
    assert torch.equal(out, want.transpose(2, 3).contiguous())
    # square_off: the squares of the outputs land behind them (one pass for x and x^2 of an LRT layer's weight gradients)
    out2 = torch.empty(2, 2, 3, 41, 37, device="cuda")
    assert ops._transpose_sum_batched(a, out2, 37, 41, (2, 3, 1), (5 * 3 * 37 * 41, 37 * 41, 0), (3 * 41 * 37, 41 * 37, 0), 41, 37,
                                      5, 3 * 37 * 41, square_off=2 * 3 * 41 * 37)
    assert torch.equal(out2[0], out) and torch.equal(out2[1], out * out)


@pytest.mark.parametrize("net_type,cin,B,lt", [("alexnet", 3, 16, "bbb"), ("3conv3fc", 3, 8, "bbb"), ("alexnet", 3, 64, "bbb"),
                                              ("alexnet", 3, 16, "lrt"), ("3conv3fc", 3, 8, "lrt"),
                                              ("lenet", 1, 8, "bbb"), ("lenet", 1, 16, "lrt")])
def test_fast_autograd_matches_reference_layout_autograd(env, net_type, cin, B, lt):
    """Same noise, same loss: every parameter gradient of the fast path equals the reference-layout path's.
    (BayesianLeNet's second conv has 6 input channels: its weight gradient pads them to 8 zero-filled planes.)"""
    ens = env["ens"]
    torch.manual_seed(1)
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, 32, 32, device="cuda")
    y = torch.randint(0, 10, (B,), device="cuda")
    E = 3
    grads = {}
    for fast in (True, False):
        ens.fast_autograd = fast
        try:
            net.zero_grad(set_to_none=True)
            env["rng"].manual_seed(5, call=7)
            lo, kl = ens.mc_forward(net, x, E, kl_mode="mean")
            assert ens.stats["path"] == ("chwn-autograd" if fast else "nchw")
            loss = F.nll_loss(lo, y) * 100.0 + 1e-6 * kl
            loss.backward()
            grads[fast] = ({n: p.grad.detach().clone() for n, p in net.named_parameters()}, lo.detach().clone(), kl.item())
        finally:
            ens.fast_autograd = True
    ga, loa, kla = grads[True]
    gb, lob, klb = grads[False]
    assert kla == klb
    np.testing.assert_allclose(loa.cpu().numpy(), lob.cpu().numpy(), rtol=2e-5, atol=2e-4)    # fused hw softplus vs torch softplus
    for n in ga:
        scale = float(gb[n].abs().max()) + 1e-12
        err = float((ga[n] - gb[n]).abs().max()) / scale
        assert err <= 2e-3, (n, err)


@pytest.mark.parametrize("net_type,cin,B,E", [("alexnet", 3, 64, 1), ("alexnet", 3, 16, 3), ("lenet", 1, 32, 1), ("3conv3fc", 3, 8, 1)])
def test_lrt_paired_backward_equals_the_two_launch_form(env, net_type, cin, B, E):
    """An LRT layer's two weight gradients (g_mu with x, g_var with x^2) and, for one draw, its two input gradients (with W_mu,
    W_var) run as the two draws of one launch each (fast_train.pair_lrt_backward): the same products as the two-launch form,
    the batch chunks of the weight gradient summed in a different grouping."""
    ens, ft = env["ens"], env["ft"]
    torch.manual_seed(4)
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, "lrt", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, 32, 32, device="cuda")
    y = torch.randint(0, 10, (B,), device="cuda")
    grads = {}
    for paired in (True, False):
        ft.pair_lrt_backward[0] = paired
        try:
            net.zero_grad(set_to_none=True)
            env["rng"].manual_seed(6, call=3)
            lo, kl = ens.mc_forward(net, x, E, kl_mode="mean")
            assert ens.stats["path"] == "chwn-autograd"
            (F.nll_loss(lo, y) * 100.0 + 1e-6 * kl).backward()
            grads[paired] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        finally:
            ft.pair_lrt_backward[0] = True
    for n in grads[True]:
        scale = float(grads[False][n].abs().max()) + 1e-12
        err = float((grads[True][n] - grads[False][n]).abs().max()) / scale
        assert err <= 2e-5, (n, err)


@pytest.mark.parametrize("E,Cout,Cin,k", [(2, 5, 7, 3), (3, 10, 128, 1), (2, 192, 64, 5), (1, 33, 50, 3), (1, 8, 4, 11), (2, 16, 16, (2, 3))])
def test_flip_transpose_w_entries(env, E, Cout, Cin, k):
    """[E, Cout, Cin, kh, kw] -> [E, Cin, Cout, kh, kw] with the taps reversed: the LDS-tiled kernel (up to 32 taps), the
    one-thread-per-output kernel (11 x 11), and the two-source form."""
    ops = env["ops"]
    kh, kw = (k, k) if isinstance(k, int) else k
    g = torch.Generator(device="cuda").manual_seed(8 + Cout)
    w0 = torch.randn(E, Cout, Cin, kh, kw, device="cuda", generator=g)
    w1 = torch.randn(E, Cout, Cin, kh, kw, device="cuda", generator=g)
    want0, want1 = (w.flip(3, 4).transpose(1, 2).contiguous() for w in (w0, w1))
    assert torch.equal(ops.flip_transpose_w(w0), want0)
    assert torch.equal(ops.flip_transpose_w_pair(w0, w1), torch.cat([want0, want1]))


def test_lenet_is_eligible_and_odd_batches_fall_back(env):
    torch.manual_seed(1)
    net = env["zoo"].getModel("lenet", 1, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    x = torch.rand(8, 1, 32, 32, device="cuda")
    assert env["ft"].train_path_ok(net, x)
    lo, kl = env["ens"].mc_forward(net, x, 2, kl_mode="mean")
    assert env["ens"].stats["path"] == "chwn-autograd" and lo.requires_grad
    x6 = torch.rand(6, 1, 32, 32, device="cuda")                  # B % 4 != 0: the reference-layout path
    assert not env["ft"].train_path_ok(net, x6)
    lo, kl = env["ens"].mc_forward(net, x6, 2, kl_mode="mean")
    assert env["ens"].stats["path"] == "nchw" and lo.requires_grad


def test_dropin_forward_with_autograd_uses_the_fast_kernels(env):
    ens = env["ens"]
    torch.manual_seed(2)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(16, 3, 32, 32, device="cuda")
    env["rng"].manual_seed(9, call=0)
    out, kl = net(x)                                            # autograd enabled: what train_model / validate_model do
    assert out.requires_grad and kl.requires_grad
    with torch.no_grad():
        env["rng"].manual_seed(9, call=0)
        out2, kl2 = net(x)
    assert torch.equal(out.detach(), out2) and kl.item() == kl2.item()
    (out.sum() + 1e-6 * kl).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("lt", ["bbb", "lrt"])
def test_alexnet_training_iterations_vs_cpu_port(env, lt):
    """3 iterations of the reference's batch loop (main_bayesian.py:40-58) on BayesianAlexNet, num_ens = 2: the GPU fast path
    (train.train_step + FusedAdam) vs the CPU port with torch.optim.Adam, both consuming the SAME Philox noise (weight noise
    for BBB layers, activation noise keyed by the NCHW output index for BBB_LRT layers)."""
    T = env["train"]
    torch.manual_seed(3)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    names = [n for n, m in net.named_children() if hasattr(m, "W_mu")]
    params = {"_prior_mu": 0, "_prior_sigma": 0.1}
    sid = {}
    for n in names:
        m = getattr(net, n)
        params[n] = {k: getattr(m, k).detach().cpu().clone() for k in ("W_mu", "W_rho", "bias_mu", "bias_rho")}
        sid[n] = m._stream_base
    g = torch.Generator().manual_seed(0)
    B, E, NB = 16, 2, 3
    batches = [(torch.rand(B, 3, 32, 32, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(NB)]
    seed, call = 31, 0
    env["rng"].manual_seed(seed, call=call)
    opt = T.FusedAdam(net.parameters(), lr=1e-3)
    losses = []
    for xb, yb in batches:
        loss, lo, kl = T.train_step(net, opt, xb.cuda(), yb.cuda(), E, 0.1, NB * B)
        assert env["ens"].stats["path"] == "chwn-autograd"
        losses.append(loss.item())
    # CPU port, same noise calls: iteration i uses calls call + i*E + j
    leaves = []
    for n in names:
        for k in ("W_mu", "W_rho", "bias_mu", "bias_rho"):
            params[n][k].requires_grad_(True)
            leaves.append(params[n][k])
    copt = torch.optim.Adam(leaves, lr=1e-3)
    KIND = {"W": 0, "bias": 1, "act": 2}
    want = []
    for i, (xb, yb) in enumerate(batches):
        copt.zero_grad()
        outs, klsum = [], 0.0
        for j in range(E):
            c = call + i * E + j
            eps = lambda name, kind, shape, c=c: torch.from_numpy(
                O.normal_eps(seed, c, sid[name] + KIND[kind], int(np.prod(shape))).reshape(shape))
            lg, k = P.forward("alexnet", params, xb, lt, "softplus", eps_fn=eps)
            outs.append(F.log_softmax(lg, dim=1))
            klsum = klsum + k
        lo = P.logmeanexp(torch.stack(outs, dim=2), 2)
        loss = F.nll_loss(lo, yb) * (NB * B) + 0.1 * (klsum / E)
        loss.backward()
        copt.step()
        want.append(loss.item())
    print(f"[fast-train alexnet {lt}] losses gpu", losses, "cpu", want)
    np.testing.assert_allclose(losses, want, rtol=2e-5)
    for n in names:
        m = getattr(net, n)
        for k in ("W_mu", "W_rho", "bias_mu", "bias_rho"):
            a, b = getattr(m, k).detach().cpu(), params[n][k].detach()
            # after 3 Adam steps each element has moved by <= 3 * lr; the two runs may disagree where |g| ~ 1e-8 (sign flips)
            assert float((a - b).abs().max()) <= 6.1e-3, (n, k)
            assert float((a - b).abs().mean()) <= 2e-5, (n, k, float((a - b).abs().mean()))


def test_training_glue_kernels_vs_torch():
    """bbb_plane_sum / bbb_sum_leading / bbb_lrt_glue (the bias gradients and LRT glue of the training backward, ATen kernels until
    round 4) against torch in float64: 1e-6 of the reduction's scale; bitwise run to run."""
    import torch
    from bbb_hip import ops
    torch.manual_seed(1)
    E, C, H, W, B = 3, 70, 5, 3, 36
    g = torch.randn(E, C, H, W, B, device="cuda")
    want = g.double().sum(dim=(2, 3, 4))
    got = ops.plane_sums(g)
    assert got.shape == (E, C) and torch.equal(got, ops.plane_sums(g))
    assert float((got.double() - want).abs().max()) <= 1e-6 * float(g.abs().sum(dim=(2, 3, 4)).max())
    assert float((ops.plane_sums(g, over_draws=True).double() - want.sum(0)).abs().max()) <= 1e-6 * float(g.abs().sum(dim=(0, 2, 3, 4)).max())
    # a padded-pitch view, as pool_act_backward_chwn(pad_planes=True) returns it
    K = H * W * B
    buf = torch.randn(E * C, ops.padded_plane_pitch(K) + 4, device="cuda")
    view = buf[:, :K].view(E, C, H, W, B)
    assert float((ops.plane_sums(view).double() - view.double().sum(dim=(2, 3, 4))).abs().max()) <= 1e-6 * K
    for shape in ((4, 6, 1, 5, 5), (5, 64, 32, 3, 3), (1, 8, 8)):            # 150 elements per draw: not a multiple of 4
        x = torch.randn(*shape, device="cuda")
        s0 = ops.sum_over_draws(x)
        assert s0.shape == tuple(shape[1:]) and float((s0.double() - x.double().sum(0)).abs().max()) <= 1e-6 * shape[0] * 4
        assert ops.sum_over_draws(x, keepdim=True).shape == (1,) + tuple(shape[1:])
    x = torch.randn(1, 16, 4, 4, 64, device="cuda")
    a, b = torch.randn(5, 16, 4, 4, 64, device="cuda"), torch.randn(5, 16, 4, 4, 64, device="cuda")
    assert torch.equal(ops.square(x), x * x)
    got = ops.lrt_input_grad_combine(a, x, b)
    want = a.double() + 2.0 * x.double() * b.double()
    assert float((got.double() - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert float((ops.lrt_input_grad_combine(a, a, b) - torch.addcmul(a, a, b, value=2.0)).abs().max()) <= 1e-5
    # ragged element counts and views with a storage offset (these used to raise BBB_ESHAPE / BBB_EALIGN in backward):
    # the scalar variant runs, same products, same fmaf
    x3 = torch.randn(512, 3, 5, 5, device="cuda")[1:]                        # 511 * 75 elements, data_ptr 300 bytes past an allocation
    assert x3.contiguous().data_ptr() % 16 != 0 or x3.numel() % 4 != 0
    assert torch.equal(ops.square(x3), x3 * x3)
    a7, b7, x7 = torch.randn(3, 7, 9, device="cuda"), torch.randn(3, 7, 9, device="cuda"), torch.randn(1, 7, 9, device="cuda")
    got = ops.lrt_input_grad_combine(a7, x7, b7)
    assert torch.equal(got, torch.addcmul(a7, (2.0 * x7).expand_as(b7), b7)) or \
        float((got.double() - (a7.double() + 2.0 * x7.double() * b7.double())).abs().max()) <= 1e-6 * 10
    off = torch.randn(4 * 64 + 1, device="cuda")[1:].view(4, 64)             # 4-byte aligned only
    assert torch.equal(ops.square(off), off * off)


@pytest.mark.parametrize("E,C,B,dev_beta", [(1, 10, 256, False), (10, 10, 512, True), (3, 100, 36, False)])
def test_fused_elbo_tail_matches_torch(E, C, B, dev_beta):
    """ops.elbo_cb_autograd = F.nll_loss(logmeanexp(log_softmax(logits)), target) * train_size + beta * kl (metrics.py:7-14,
    main_bayesian.py:49-56 upstream), values and both gradients; beta as a Python number or a device scalar (a captured step)."""
    from bbb_hip import ops
    g = torch.Generator(device="cuda").manual_seed(E * 100 + C)
    logits = (torch.randn(E, C, B, device="cuda", generator=g) * 3).requires_grad_(True)
    kl = (torch.rand((), device="cuda", generator=g) * 1e4).requires_grad_(True)
    target = torch.randint(0, C, (B,), device="cuda", generator=g)
    beta, n = 0.037, 50000.0
    bt = torch.full((), beta, device="cuda") if dev_beta else beta
    loss, lse = ops.elbo_cb_autograd(logits, kl, target, bt, n, E)
    assert not lse.requires_grad
    loss.backward()
    l2 = logits.detach().double().requires_grad_(True)
    k2 = kl.detach().double().requires_grad_(True)
    lse2 = torch.logsumexp(F.log_softmax(l2.permute(0, 2, 1), dim=2), dim=0) - np.log(E)
    want = F.nll_loss(lse2, target) * n + beta * k2
    want.backward()
    np.testing.assert_allclose(lse.cpu().numpy(), lse2.detach().float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    assert abs(loss.item() - want.item()) <= 2e-6 * abs(want.item())
    np.testing.assert_allclose(logits.grad.cpu().numpy(), l2.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-6 * n / B)
    assert abs(kl.grad.item() - beta) <= 1e-7


def test_hip_loss_tail_matches_the_torch_tail():
    """ops.mc_tail_cb_autograd (log_softmax + logmeanexp over the draws, forward and backward one HIP launch each) against the torch
    ops it replaced in the training step: values to 2e-6, gradients to 1e-5 of their largest magnitude, for 10 / 100 classes, one and
    several draws, mean_over 0 and E; float64 autograd as the judge of both."""
    import math
    import torch.nn.functional as F
    from bbb_hip import ops
    for E, C, B, mo in ((10, 10, 512, 10), (1, 100, 256, 1), (3, 7, 36, 0)):
        torch.manual_seed(E + C)
        lg = (torch.randn(E, C, B, device="cuda") * 5).requires_grad_(True)
        y = torch.randint(0, C, (B,), device="cuda")
        w = torch.randn(B, C, device="cuda")

        def torch_tail(t):
            return torch.logsumexp(F.log_softmax(t.permute(0, 2, 1), dim=2), dim=0) - (math.log(mo) if mo > 0 else 0.0)

        a = ops.mc_tail_cb_autograd(lg, mo)
        b = torch_tail(lg)
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
        for loss_of in (lambda t: F.nll_loss(t, y) * 50000.0, lambda t: (t * w).sum()):
            ga, = torch.autograd.grad(loss_of(a), lg, retain_graph=True)
            gb, = torch.autograd.grad(loss_of(b), lg, retain_graph=True)
            g64, = torch.autograd.grad(loss_of(torch_tail(lg.double())), lg, retain_graph=True)
            scale = float(g64.abs().max())
            assert float((ga - g64).abs().max()) <= 1e-5 * scale and float((gb - g64).abs().max()) <= 1e-5 * scale


def test_flip_transpose_w_multi_entry(env):
    """Several layers' input-gradient weights (single sets and mean / variance pairs) flipped in one launch."""
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(77)
    a = torch.randn(3, 12, 8, 3, 3, device="cuda", generator=g)
    b0, b1 = (torch.randn(1, 10, 128, 1, 1, device="cuda", generator=g) for _ in range(2))
    c = torch.randn(2, 5, 7, 5, 5, device="cuda", generator=g)
    outs = ops.flip_transpose_w_multi([a, (b0, b1), c])
    assert torch.equal(outs[0], ops.flip_transpose_w(a))
    assert torch.equal(outs[1], ops.flip_transpose_w_pair(b0, b1))
    assert torch.equal(outs[2], ops.flip_transpose_w(c))
    many = [torch.randn(1, 4, 4, 1, 1, device="cuda", generator=g) for _ in range(19)]          # more than one launch's 16 segments
    for o, w in zip(ops.flip_transpose_w_multi(many), many):
        assert torch.equal(o, ops.flip_transpose_w(w))


@pytest.mark.parametrize("net_type,cin,B,E", [("alexnet", 3, 32, 1), ("3conv3fc", 3, 8, 2), ("lenet", 1, 16, 3)])
def test_lrt_combine_folded_into_the_layer_below(env, net_type, cin, B, E):
    """g1 + 2 x g2 (the two input gradients of an LRT layer) formed inside the pooling / activation backward of the layer below
    (bbb_lrt_pool_act_bwd_chwn g_out2 / x_out) = the separate bbb_lrt_glue launch, bit for bit: same fmaf per element."""
    ens, ft = env["ens"], env["ft"]
    torch.manual_seed(9)
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, "lrt", "relu" if net_type == "lenet" else "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, 32, 32, device="cuda")
    y = torch.randint(0, 10, (B,), device="cuda")
    grads = {}
    for fold in (True, False):
        ft.fold_lrt_combine[0] = fold
        try:
            net.zero_grad(set_to_none=True)
            env["rng"].manual_seed(8, call=1)
            lo, kl = ens.mc_forward(net, x, E, kl_mode="mean")
            assert ens.stats["path"] == "chwn-autograd"
            (F.nll_loss(lo, y) * 100.0 + 1e-6 * kl).backward()
            grads[fold] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        finally:
            ft.fold_lrt_combine[0] = True
    for n in grads[True]:
        assert torch.equal(grads[True][n], grads[False][n]), n
