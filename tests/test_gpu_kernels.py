"""HIP kernels (through the C ABI / ctypes) against the numpy oracle and the reference-generated fixtures.
Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch

import bbb_numpy as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from bbb_hip import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


# ---------------------------------------------------------------- noise stream
def test_eps_matches_oracle_stream(ops):
    seed, call, stream = 0x1234567890ABCDEF, 7, 13
    got = ops.eps_dump(100003, seed, call, stream, "cuda").cpu().numpy()
    want = O.normal_eps(seed, call, stream, 100003)
    # hardware log2 / sin / cos vs float64 libm: stated tolerance 2e-5 absolute on N(0,1) samples
    assert np.max(np.abs(got - want)) < 2e-5
    off = ops.eps_dump(1001, seed, call, stream, "cuda", start=4099).cpu().numpy()
    np.testing.assert_array_equal(off, got[4099:5100])


def test_eps_moments(ops):
    z = ops.eps_dump(1 << 22, 99, 0, 1, "cuda").double()
    assert abs(z.mean().item()) < 2e-3 and abs(z.std().item() - 1) < 2e-3
    assert abs((z ** 3).mean().item()) < 1e-2 and abs((z ** 4).mean().item() - 3) < 3e-2
    assert torch.isfinite(z).all()


# ---------------------------------------------------------------- fused reparam + KL
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 1024, 1025, 40003])
def test_reparam_kl_ragged_sizes(ops, n):
    rng = np.random.default_rng(n)
    mu = (rng.standard_normal(n) * 0.1).astype(np.float32)
    rho = (rng.standard_normal(n) * 0.1 - 5).astype(np.float32)
    eps = rng.standard_normal((3, n)).astype(np.float32)
    ws, sig, kl = ops.reparam_kl_forward([dev(mu)], [dev(rho)], 0.0, 0.1, [5], 1, 0, draws=3, want_sigma=True,
                                         eps=[dev(eps)])
    s = O.sigma_from_rho(rho)
    np.testing.assert_allclose(sig[0].cpu().numpy(), s, rtol=5e-7)     # <= 4 ulp (hardware exp2 + atanh series)
    np.testing.assert_allclose(ws[0].cpu().numpy(), mu[None] + eps * s[None], rtol=1e-6, atol=1e-7)
    want = O.kl_loss(mu, s, 0.0, 0.1)
    assert abs(kl.item() - want) <= 1e-6 * abs(want)


def test_reparam_kl_multitensor_philox_and_determinism(ops):
    rng = np.random.default_rng(0)
    shapes = [(64, 3, 11, 11), (64,), (192, 64, 5, 5), (192,), (10, 128), (10,), (7,)]
    mus = [(rng.standard_normal(s) * 0.1).astype(np.float32) for s in shapes]
    rhos = [(rng.standard_normal(s) * 0.1 - 5).astype(np.float32) for s in shapes]
    ids = list(range(len(shapes)))
    seed, call0, E = 42, 3, 4
    dm, dr = [dev(m) for m in mus], [dev(r) for r in rhos]
    ws, _, kl = ops.reparam_kl_forward(dm, dr, 0.0, 0.1, ids, seed, call0, draws=E)
    ws2, _, kl2 = ops.reparam_kl_forward(dm, dr, 0.0, 0.1, ids, seed, call0, draws=E)
    assert kl.item() == kl2.item()                        # fixed reduction tree: bitwise reproducible
    want_kl = sum(O.kl_loss(m, O.sigma_from_rho(r), 0.0, 0.1) for m, r in zip(mus, rhos))
    assert abs(kl.item() - want_kl) <= 1e-6 * want_kl
    for i, (m, r) in enumerate(zip(mus, rhos)):
        assert torch.equal(ws[i], ws2[i])
        s = O.sigma_from_rho(r).reshape(-1)
        for e in range(E):
            eps = O.normal_eps(seed, call0 + e, ids[i], m.size)
            want = m.reshape(-1) + eps * s
            np.testing.assert_allclose(ws[i][e].cpu().numpy().reshape(-1), want, rtol=1e-6, atol=2e-7)
    # draw e of a batched launch == a single-draw launch at call0 + e
    w1, _, _ = ops.reparam_kl_forward(dm, dr, 0.0, 0.1, ids, seed, call0 + 2, draws=1)
    for i in range(len(shapes)):
        assert torch.equal(w1[i][0], ws[i][2])


def test_reparam_kl_against_reference_fixture(ops, golden):
    Fn = golden["functions"]
    _, sig, kl = ops.reparam_kl_forward([dev(Fn["kl.mu"])], [dev(Fn["kl.rho"])], 0, 0.1, [0], 0, 0, sample=False, want_sigma=True)
    np.testing.assert_allclose(sig[0].cpu().numpy(), Fn["kl.sigma"], rtol=5e-7)
    assert abs(kl.item() - float(Fn["kl.value_cfg"])) <= 2e-6 * float(Fn["kl.value_cfg"])
    _, _, klt = ops.reparam_kl_forward([dev(Fn["kl.mu"])], [dev(Fn["kl.rho"])], 0, 0.1, [0], 0, 0, sample=False, textbook_kl=True)
    assert abs(klt.item() - float(Fn["kl.value_textbook"])) <= 2e-6 * float(Fn["kl.value_textbook"])
    _, s2, _ = ops.reparam_kl_forward([dev(Fn["kl.mu"])], [dev(Fn["kl.rho"])], 0, 0.1, [0], 0, 0, sample=False, want_sigma=True,
                                      sigma_squared=True, want_kl=False)
    np.testing.assert_allclose(s2[0].cpu().numpy(), Fn["kl.sigma"] ** 2, rtol=1e-6)


def test_reparam_large_rho_is_finite(ops):
    mu = torch.zeros(8, device="cuda")
    rho = torch.tensor([-30., -10., 0., 10., 19.9, 20.1, 60., 100.], device="cuda")
    _, sig, _ = ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 0, 0, sample=False, want_sigma=True)
    with np.errstate(over="ignore"):
        want = O.sigma_from_rho(rho.cpu().numpy())
    got = sig[0].cpu().numpy()
    np.testing.assert_allclose(got[:7], want[:7], rtol=5e-7)
    assert got[7] == 100.0 and np.isinf(want[7])       # documented departure: no overflow


def test_reparam_backward_matches_oracle_grads(ops):
    rng = np.random.default_rng(3)
    n, E = 777, 3
    mu = (rng.standard_normal(n) * 0.1).astype(np.float32)
    rho = (rng.standard_normal(n) * 0.3 - 3).astype(np.float32)
    gw = rng.standard_normal((E, n)).astype(np.float32)
    seed, call0, sid = 5, 11, 2
    gkl = torch.tensor(0.37, device="cuda")
    gm, gr = ops.reparam_kl_backward([dev(mu)], [dev(rho)], [dev(gw)], gkl, 0.0, 0.1, [sid], seed, call0, E)
    eps = np.stack([O.normal_eps(seed, call0 + e, sid, n) for e in range(E)]).astype(np.float64)
    kmu, krho = O.kl_grads(mu, rho, 0.0, 0.1)
    sgm = 1 / (1 + np.exp(-rho.astype(np.float64)))
    np.testing.assert_allclose(gm[0].cpu().numpy(), gw.sum(0) + 0.37 * kmu, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gr[0].cpu().numpy(), (gw * eps).sum(0) * sgm + 0.37 * krho, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("squared", [False, True])
def test_reparam_backward_takes_the_sigma_output_gradient(ops, squared):
    """The forward's sigma / sigma^2 output (the LRT layers' variance operand, layers/BBB_LRT/BBBConv.py:64-69) is
    differentiable: its incoming gradient enters grad_rho through d sigma / d rho = sigmoid(rho) (x 2 sigma when squared),
    inside bbb_reparam_kl_bwd; checked against torch autograd in float64 through ops.kl_only."""
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(6, 5, 3, 3), (6,), (10, 7)]
    mus = [(torch.randn(sh, device="cuda", generator=g) * 0.1).requires_grad_(True) for sh in shapes]
    rhos = [(torch.randn(sh, device="cuda", generator=g) * 0.5 - 3).requires_grad_(True) for sh in shapes]
    gs = [torch.randn(sh, device="cuda", generator=g) for sh in shapes]
    kl, sig = ops.kl_only(mus, rhos, 0.0, 0.1, want_sigma=True, sigma_squared=squared)
    loss = 0.37 * kl + sum((a * b).sum() for a, b in zip(sig[:2], gs[:2]))          # the third sigma output gets no gradient
    loss.backward()
    mus64 = [m.detach().double().requires_grad_(True) for m in mus]
    rhos64 = [r.detach().double().requires_grad_(True) for r in rhos]
    ref = 0.0
    for i, (m, r) in enumerate(zip(mus64, rhos64)):
        sg = torch.log1p(torch.exp(r))
        ref = ref + 0.37 * 0.5 * (2 * torch.log(sg / 0.1) - 1 + (0.1 / sg) ** 2 + ((m - 0.0) / sg) ** 2).sum()   # metrics.py:28
        if i < 2:
            ref = ref + ((sg * sg if squared else sg) * gs[i].double()).sum()
    ref.backward()
    for m, m64, r, r64 in zip(mus, mus64, rhos, rhos64):
        np.testing.assert_allclose(m.grad.cpu().numpy(), m64.grad.float().cpu().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(r.grad.cpu().numpy(), r64.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)


# ---------------------------------------------------------------- conv / linear on the fp32 matrix cores
CONV_CASES = [
    # B, Cin, H, W, Cout, kh, kw, stride, pad, dil, E, x_shared
    (2, 3, 9, 9, 5, 3, 3, 2, 1, 1, 1, False),
    (3, 2, 8, 7, 4, 2, 3, 1, 2, 2, 1, False),       # ragged kernel, dilation
    (5, 3, 32, 32, 64, 11, 11, 4, 5, 1, 2, True),   # AlexNet conv1 geometry, K = 363 (not a multiple of 4)
    (4, 64, 4, 4, 192, 5, 5, 1, 2, 1, 3, False),    # AlexNet conv2
    (7, 192, 2, 2, 384, 3, 3, 1, 1, 1, 2, False),   # AlexNet conv3 (2x2 images)
    (6, 1, 32, 32, 6, 5, 5, 1, 0, 1, 1, False),     # LeNet conv1 (Cout = 6 << tile)
    (3, 32, 15, 15, 64, 5, 5, 1, 2, 1, 1, False),   # 3Conv3FC conv2
    (130, 16, 6, 6, 70, 3, 3, 1, 1, 1, 1, False),   # M and Cout straddle tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_against_oracle(ops, case):
    B, Cin, H, W, Cout, kh, kw, s, p, d, E, shared = case
    rng = np.random.default_rng(sum(case[:11]))
    x = rng.standard_normal((1 if shared else E, B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((E, Cout, Cin, kh, kw)) * 0.1).astype(np.float32)
    b = rng.standard_normal((E, Cout)).astype(np.float32)
    y = ops.conv2d_forward(dev(x), dev(w), dev(b), s, p, d).cpu().numpy()
    for e in range(E):
        want = O.conv2d(x[0 if shared else e], w[e], b[e], s, p, d)
        # fp32 fmaf chain vs fp64 accumulation: ~1e-7 * sum|a*b|; stated as rtol/atol 2e-5 on O(1) outputs
        np.testing.assert_allclose(y[e], want, rtol=2e-5, atol=2e-5)


def test_conv2d_transpose_detecting(ops):
    """Asymmetric operands: a swapped row/col or (kh,kw) mapping cannot pass."""
    B, Cin, H, W, Cout = 1, 2, 5, 6, 3
    x = np.arange(B * Cin * H * W, dtype=np.float32).reshape(1, B, Cin, H, W) / 10
    w = np.zeros((1, Cout, Cin, 2, 3), np.float32)
    w[0, 1, 0, 0, 2] = 1.0
    w[0, 2, 1, 1, 0] = -2.0
    y = ops.conv2d_forward(dev(x), dev(w), None, 1, 0, 1).cpu().numpy()[0]
    np.testing.assert_allclose(y, O.conv2d(x[0], w[0], None, 1, 0, 1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M,K,N,E", [(5, 7, 4, 1), (64, 400, 120, 2), (256, 512, 1000, 1), (512, 128, 10, 3), (33, 84, 10, 1)])
def test_linear_against_oracle(ops, M, K, N, E):
    rng = np.random.default_rng(M * K + N)
    x = rng.standard_normal((E, M, K)).astype(np.float32)
    w = (rng.standard_normal((E, N, K)) * 0.1).astype(np.float32)
    b = rng.standard_normal((E, N)).astype(np.float32)
    y = ops.conv2d_forward(dev(x).reshape(E, M, K, 1, 1), dev(w).reshape(E, N, K, 1, 1), dev(b)).cpu().numpy().reshape(E, M, N)
    for e in range(E):
        np.testing.assert_allclose(y[e], O.linear(x[e], w[e], b[e]), rtol=2e-5, atol=2e-5)


def test_conv2d_fused_activation(ops):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 4, 3, 8, 8)).astype(np.float32) * 5
    w = rng.standard_normal((1, 6, 3, 3, 3)).astype(np.float32)
    base = O.conv2d(x[0], w[0], None, 1, 1, 1)
    np.testing.assert_allclose(ops.conv2d_forward(dev(x), dev(w), None, 1, 1, 1, act="relu").cpu().numpy()[0], O.relu_act(base), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(ops.conv2d_forward(dev(x), dev(w), None, 1, 1, 1, act="softplus").cpu().numpy()[0], O.softplus_act(base), rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------- LRT dual-accumulator kernel
@pytest.mark.parametrize("case", [(3, 4, 6, 6, 6, 3, 1, 1, 2), (2, 3, 32, 32, 64, 11, 4, 5, 1), (5, 64, 4, 4, 192, 5, 1, 2, 2), (9, 40, 1, 1, 10, 1, 1, 0, 1)])
def test_lrt_moments_and_output(ops, case):
    B, Cin, H, W, Cout, k, s, p, E = case
    rng = np.random.default_rng(sum(case))
    x = rng.random((E, B, Cin, H, W)).astype(np.float32)
    wmu = (rng.standard_normal((Cout, Cin, k, k)) * 0.1).astype(np.float32)
    wrho = (rng.standard_normal((Cout, Cin, k, k)) * 0.1 - 3).astype(np.float32)
    bmu = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    brho = (rng.standard_normal(Cout) * 0.1 - 3).astype(np.float32)
    wvar = O.sigma_from_rho(wrho) ** 2
    bvar = O.sigma_from_rho(brho) ** 2
    seed, call0, sid = 77, 5, 6
    y, am, av = ops.lrt_conv2d_forward(dev(x), dev(wmu), dev(wvar), dev(bmu), dev(bvar), seed, call0, sid, s, p, 1, want_moments=True)
    for e in range(E):
        wam, wav = O.lrt_moments_conv2d(x[e], wmu, wrho, bmu, brho, s, p, 1)
        np.testing.assert_allclose(am[e].cpu().numpy(), wam, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(av[e].cpu().numpy(), wav, rtol=3e-5, atol=1e-9)
        eps = O.normal_eps(seed, call0 + e, sid, wam.size).reshape(wam.shape)
        np.testing.assert_allclose(y[e].cpu().numpy(), O.lrt_output(wam, wav, eps), rtol=3e-5, atol=3e-5)
    # external eps (replay entry) and sample=False
    eps = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y2, _, _ = ops.lrt_conv2d_forward(dev(x), dev(wmu), dev(wvar), dev(bmu), dev(bvar), 0, 0, 0, s, p, 1, eps=dev(eps))
    y3, _, _ = ops.lrt_conv2d_forward(dev(x), dev(wmu), dev(wvar), dev(bmu), dev(bvar), 0, 0, 0, s, p, 1, sample=False)
    for e in range(E):
        wam, wav = O.lrt_moments_conv2d(x[e], wmu, wrho, bmu, brho, s, p, 1)
        np.testing.assert_allclose(y2[e].cpu().numpy(), O.lrt_output(wam, wav, eps[e]), rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(y3[e].cpu().numpy(), wam, rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------- Monte-Carlo tail
@pytest.mark.parametrize("E,B,C", [(1, 4, 10), (10, 512, 10), (3, 7, 100), (25, 16, 257)])
def test_mc_tail(ops, E, B, C):
    rng = np.random.default_rng(E * B + C)
    z = (rng.standard_normal((E, B, C)) * 4).astype(np.float32)
    got = ops.mc_tail(dev(z), mean_over=E).cpu().numpy()
    np.testing.assert_allclose(got, O.mc_log_outputs(z), rtol=2e-6, atol=2e-6)
    lse = ops.mc_tail(dev(z), mean_over=0).cpu().numpy()
    np.testing.assert_allclose(lse - np.log(E), got, rtol=2e-6, atol=2e-6)


def test_mc_tail_against_reference_fixture(ops, golden):
    M = golden["models"]
    got = ops.mc_tail(dev(M["mc_lenet.logits"]), mean_over=3).cpu().numpy()
    np.testing.assert_allclose(got, M["mc_lenet.log_outputs"], rtol=1e-5, atol=2e-6)


# ---------------------------------------------------------------- argument errors
def test_errors_are_loud(ops):
    from bbb_hip import BBBHipError
    with pytest.raises(BBBHipError):
        ops.conv2d_forward(torch.zeros(1, 1, 3, 4, 4), torch.zeros(1, 2, 3, 3, 3), None)          # CPU tensors
    with pytest.raises(BBBHipError):
        ops.conv2d_forward(torch.zeros(1, 1, 3, 4, 4).cuda(), torch.zeros(1, 2, 4, 3, 3).cuda(), None)  # channel mismatch
    with pytest.raises(BBBHipError):
        ops.conv2d_forward(torch.zeros(1, 1, 3, 2, 2).cuda(), torch.zeros(1, 2, 3, 3, 3).cuda(), None)  # kernel > image
    with pytest.raises(BBBHipError):
        ops.reparam_kl_forward([torch.zeros(4).cuda()], [torch.zeros(4).cuda()], 0, -1.0, [0], 0, 0)     # bad prior sigma


# ---------------------------------------------------------------- batch-innermost (CHWN) kernels of the ensemble path
CHWN_CASES = [
    # B, Cin, H, W, Cout, kh, kw, stride, pad, dil, E, x_shared
    (8, 3, 9, 9, 5, 3, 3, 2, 1, 1, 1, False),
    (4, 2, 8, 7, 4, 2, 3, 1, 2, 2, 2, False),
    (132, 3, 32, 32, 64, 11, 11, 4, 5, 1, 2, True),    # AlexNet conv1; B straddles the 128 tile
    (64, 64, 4, 4, 192, 5, 5, 1, 2, 1, 3, False),      # AlexNet conv2 (half the taps are padding)
    (260, 192, 2, 2, 384, 3, 3, 1, 1, 1, 2, False),    # AlexNet conv3 (2x2 maps)
    (12, 1, 32, 32, 6, 5, 5, 1, 0, 1, 1, False),       # LeNet conv1
    (8, 32, 15, 15, 64, 5, 5, 1, 2, 1, 1, False),      # 3Conv3FC conv2
    (36, 16, 6, 6, 70, 3, 3, 1, 1, 1, 1, False),
    (256, 512, 1, 1, 1000, 1, 1, 1, 0, 1, 1, False),   # linear as 1x1
    (4, 5, 3, 3, 7, 5, 5, 1, 3, 1, 1, False),          # kernel larger than the image: most taps out of bounds
]


@pytest.mark.parametrize("case", CHWN_CASES)
def test_conv2d_chwn_matches_oracle_and_nchw_kernel(ops, case):
    B, Cin, H, W, Cout, kh, kw, s, p, d, E, shared = case
    rng = np.random.default_rng(sum(case[:11]) + 1)
    x = rng.standard_normal((1 if shared else E, B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((E, Cout, Cin, kh, kw)) * 0.1).astype(np.float32)
    b = rng.standard_normal((E, Cout)).astype(np.float32)
    xd = dev(x)
    y = ops.conv2d_chwn_forward(xd.permute(0, 2, 3, 4, 1).contiguous(), dev(w), dev(b), s, p, d)      # [E, Cout, Ho, Wo, B]
    y = y.permute(0, 4, 1, 2, 3).contiguous()
    y_nchw = ops.conv2d_forward(xd, dev(w), dev(b), s, p, d)
    # skipping padding taps only drops exact zeros from the fmaf chain: same bits as the NCHW kernel (small launches split
    # their contraction over several workgroups by default, which rounds the partial sums separately: compared unsplit)
    saved, ops.split_k = ops.split_k, False
    try:
        y_unsplit = ops.conv2d_chwn_forward(xd.permute(0, 2, 3, 4, 1).contiguous(), dev(w), dev(b), s, p, d).permute(0, 4, 1, 2, 3).contiguous()
    finally:
        ops.split_k = saved
    assert torch.equal(y_unsplit, y_nchw)
    assert float((y - y_unsplit).abs().max()) <= 4e-6 * max(1.0, float(y_unsplit.abs().max()))
    for e in range(E):
        np.testing.assert_allclose(y[e].cpu().numpy(), O.conv2d(x[0 if shared else e], w[e], b[e], s, p, d), rtol=2e-5, atol=2e-5)


def test_lrt_chwn_matches_nchw_kernel_and_oracle(ops):
    B, Cin, H, W, Cout, k, s, p, E = 8, 16, 6, 6, 70, 3, 1, 1, 2
    rng = np.random.default_rng(5)
    x = rng.random((E, B, Cin, H, W)).astype(np.float32)
    wmu = (rng.standard_normal((Cout, Cin, k, k)) * 0.1).astype(np.float32)
    wvar = (rng.random((Cout, Cin, k, k)) * 1e-3).astype(np.float32)
    bmu = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    bvar = (rng.random(Cout) * 1e-3).astype(np.float32)
    xd = dev(x)
    y1, am1, av1 = ops.lrt_conv2d_forward(xd, dev(wmu), dev(wvar), dev(bmu), dev(bvar), 7, 3, 2, s, p, 1, want_moments=True, act="softplus")
    y2, am2, av2 = ops.lrt_conv2d_chwn_forward(xd.permute(0, 2, 3, 4, 1).contiguous(), dev(wmu), dev(wvar), dev(bmu), dev(bvar),
                                               7, 3, 2, s, p, 1, want_moments=True, act="softplus")
    back = lambda t: t.permute(0, 4, 1, 2, 3).contiguous()
    assert torch.equal(back(am2), am1) and torch.equal(back(av2), av1)
    assert torch.equal(back(y2), y1)                           # same eps stream (canonical NCHW element index)
    for e in range(E):
        want = O.conv2d(x[e], wmu, bmu, s, p, 1)
        np.testing.assert_allclose(am1[e].cpu().numpy(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("k,s,H,W", [(2, 2, 8, 8), (3, 2, 15, 15), (3, 2, 7, 9), (2, 2, 5, 5)])
def test_maxpool_chwn(ops, k, s, H, W):
    rng = np.random.default_rng(k * H)
    x = rng.standard_normal((3, 5, H, W, 8)).astype(np.float32)
    y = ops.maxpool_chwn(dev(x), k, s).cpu().numpy()
    want = O.maxpool2d(np.ascontiguousarray(x.transpose(0, 4, 1, 2, 3)).reshape(-1, 5, H, W), k, s)
    want = want.reshape(3, 8, 5, *want.shape[2:]).transpose(0, 2, 3, 4, 1)
    np.testing.assert_array_equal(y, want)


@pytest.mark.parametrize("E,B,C", [(1, 4, 10), (10, 512, 10), (3, 70, 100), (80, 16, 7)])
def test_mc_tail_batch_innermost(ops, E, B, C):
    rng = np.random.default_rng(E + B + C)
    z = (rng.standard_normal((E, B, C)) * 4).astype(np.float32)
    zcb = dev(np.ascontiguousarray(z.transpose(0, 2, 1)))
    np.testing.assert_allclose(ops.mc_tail_cb(zcb, mean_over=E).cpu().numpy(), O.mc_log_outputs(z), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ops.mc_tail_cb(zcb).cpu().numpy(), ops.mc_tail(dev(z)).cpu().numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("shape", [(512, 3, 32, 32), (4, 1, 5, 7), (33, 2, 3, 3)])
def test_to_batch_innermost(ops, shape):
    x = torch.randn(*shape, device="cuda")
    assert torch.equal(ops.to_batch_innermost(x), x.permute(1, 2, 3, 0).contiguous())


def test_layout_transposes_of_a_large_feature_map(ops):
    """More than 65535 x 32 elements per image (64 channels x 224 x 224: the per-layer path of a 224 x 224 model): the row-tile
    count used to sit on gridDim.y and the call failed with BBB_ESHAPE."""
    x = torch.randn(4, 64, 224, 224, device="cuda")
    t = ops.to_batch_innermost(x)
    assert torch.equal(t, x.permute(1, 2, 3, 0).contiguous())
    assert torch.equal(ops.from_batch_innermost(t), x)
    w = torch.randn(1, 8, 64, 3, 3, device="cuda") * 0.05
    with torch.no_grad():
        y = ops.conv2d_layer(x, w, None, 1, 1, 1)                      # the per-layer fast path: transpose, conv, transpose back
    ref = torch.nn.functional.conv2d(x.double(), w[0].double(), None, 1, 1, 1)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 2e-5


# ---------------------------------------------------------------- backward on the same GEMM kernel (training extension)
@pytest.mark.parametrize("case", [
    # B, Cin, H, W, Cout, kh, kw, stride, pad, dil, E, w_shared
    (6, 5, 9, 8, 7, 3, 3, 1, 1, 1, 2, False),
    (4, 3, 12, 12, 6, 5, 5, 2, 2, 1, 1, False),      # strided: wgrad via dilation = stride (with cropping), dgrad via upsampling
    (5, 4, 10, 9, 3, 3, 2, 3, 1, 1, 2, False),       # stride 3, (H + 2p - k) % s != 0
    (8, 16, 4, 4, 12, 3, 3, 1, 1, 1, 3, True),       # weights shared by the draws (LRT-style): grads summed over draws
    (4, 6, 7, 7, 5, 3, 3, 1, 0, 2, 1, False),        # dilated
    (16, 40, 1, 1, 10, 1, 1, 1, 0, 1, 2, False),     # linear as 1x1
    (4, 3, 11, 9, 5, 3, 3, 2, 3, 1, 2, False),       # padding larger than the kernel reach (p > k - 1), strided
    (4, 4, 13, 13, 6, 3, 3, 2, 1, 2, 1, False),      # strided AND dilated
])
def test_conv_backward_helpers_vs_autograd(ops, case):
    B, Cin, H, W, Cout, kh, kw, s, p, d, E, shared = case
    g = torch.Generator(device="cuda").manual_seed(sum(case[:11]))
    x = torch.randn(E, B, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(1 if shared else E, Cout, Cin, kh, kw, device="cuda", generator=g) * 0.2
    ys = [torch.nn.functional.conv2d(x[e], w[0 if shared else e], None, s, p, d) for e in range(E)]
    gy = torch.randn(E, *ys[0].shape, device="cuda", generator=g)
    # reference gradients from ATen in float64
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    loss = sum((torch.nn.functional.conv2d(xd[e], wd[0 if shared else e], None, s, p, d) * gy[e].double()).sum() for e in range(E))
    loss.backward()
    gw = ops.conv2d_weight_grad(gy, x, tuple(w.shape), s, p, d)
    np.testing.assert_allclose(gw.cpu().numpy(), wd.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)
    gx = ops.conv2d_input_grad(gy, w, tuple(x.shape), s, p, d)       # strided: zero-upsampled gy; dilated: dilated flipped kernel
    np.testing.assert_allclose(gx.cpu().numpy(), xd.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)
    # through the autograd Function (bias included)
    xa = x.clone().requires_grad_(True)
    wa = w.clone().requires_grad_(True)
    ba = torch.randn(w.shape[0], Cout, device="cuda", generator=g).requires_grad_(True)
    (ops.conv2d(xa, wa, ba, s, p, d) * gy).sum().backward()
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xd.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(wa.grad.cpu().numpy(), wd.grad.float().cpu().numpy(), rtol=2e-4, atol=2e-4)
    want_gb = gy.sum(dim=(1, 3, 4))
    np.testing.assert_allclose(ba.grad.cpu().numpy(), (want_gb.sum(0, keepdim=True) if shared else want_gb).cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p", [(16, 3, 32, 32, 64, 11, 4, 5), (8, 2, 7, 5, 10, 3, 1, 1), (12, 4, 6, 6, 5, 3, 2, 0)])
def test_lrt_shared_input_dedup_is_bitwise_the_full_launch(ops, B, Cin, H, W, Cout, k, s, p):
    """First LRT layer of an ensemble: x and (mu, sigma^2) are the same for all draws, so the moments are computed once
    and bbb_lrt_sample_chwn draws the E outputs -- must equal E full LRT launches bit for bit (pixels % 4 == 0 and != 0)."""
    torch.manual_seed(B + Cout)
    E, seed, call0, sid = 5, 321, 9, 6
    x = torch.rand(1, Cin, H, W, B, device="cuda")
    w_mu = torch.randn(Cout, Cin, k, k, device="cuda") * 0.2
    w_var = torch.rand(Cout, Cin, k, k, device="cuda") * 0.01
    b_mu, b_var = torch.randn(Cout, device="cuda"), torch.rand(Cout, device="cuda") * 0.01
    full, _, _ = ops.lrt_conv2d_chwn_forward(x.expand(E, -1, -1, -1, -1).contiguous(), w_mu, w_var, b_mu, b_var, seed, call0, sid,
                                             s, p, 1, sample=True, act="softplus")
    _, am, av = ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, seed, call0, sid, s, p, 1, sample=False,
                                            want_moments=True, act=None)
    got = ops.lrt_sample_chwn(am, av, E, seed, call0, sid, act="softplus")
    assert got.shape == full.shape
    assert torch.equal(got, full)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,E,shared_w", [(256, 384, 2, 2, 256, 3, 1, 1, 1, False), (32, 64, 4, 4, 192, 5, 1, 2, 2, True),
                                                             (16, 24, 6, 6, 10, 3, 2, 1, 1, False)])
def test_conv2d_splitk_matches_the_single_launch(ops, B, Cin, H, W, Cout, k, s, p, E, shared_w):
    """Training-path split-K (input channels as extra draws of one launch, fixed-order sum) = the plain launch up to fp32
    summation order; falls through to the plain launch when the channels cannot be split."""
    torch.manual_seed(Cin + B)
    x = torch.randn(E, B, Cin, H, W, device="cuda")
    w = torch.randn(1 if shared_w else E, Cout, Cin, k, k, device="cuda") * 0.1
    b = torch.randn(w.shape[0], Cout, device="cuda")
    want = ops.conv2d_forward(x, w, b, s, p, 1)
    got = ops.conv2d_splitk(x, w, b, s, p, 1)
    assert got.shape == want.shape
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale
    assert torch.equal(ops.conv2d_splitk(x, w, b, s, p, 1), got)            # deterministic
