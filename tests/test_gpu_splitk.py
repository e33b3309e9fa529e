"""Split contraction of small batch-innermost launches (bbb_conv2d_chwn_splitk_fwd / bbb_lrt_conv2d_chwn_splitk_fwd, on by
default through ops.split_k): an output tile's k range is cut over several workgroups, partial tiles are added in range
order by the last arriver.  Stated bounds: split vs unsplit launches differ by at most 4e-6 of max|output| (partial sums of
up to 1536 terms rounded separately; measured 2.3e-6 on the models below); the oracle bound of the unsplit kernel (2e-5)
holds unchanged; a split launch is bitwise reproducible run to run; differently sized launches share one scratch buffer."""
import ctypes

import numpy as np
import pytest
import torch

import bbb_numpy as O

pytestmark = pytest.mark.gpu
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
TOL = 4e-6

# AlexNet / CIFAR layers at one draw, bs 512 (B, Cin, H, W, Cout, k, stride, pad) -> expected k ranges
LAYERS = [("conv1", 512, 3, 32, 32, 64, 11, 4, 5, 1), ("conv2", 512, 64, 4, 4, 192, 5, 1, 2, 2), ("conv3", 512, 192, 2, 2, 384, 3, 1, 1, 2),
          ("conv4", 512, 384, 2, 2, 256, 3, 1, 1, 4), ("conv5", 512, 256, 2, 2, 128, 3, 1, 1, 4), ("fc", 512, 128, 1, 1, 10, 1, 1, 0, 1)]


def _plan(B, Cin, H, W, Cout, k, st, pd, E, lrt=False):
    from bbb_hip import _lib
    d = _lib.ConvDesc()
    d.batch, d.cin, d.h, d.w, d.cout, d.kh, d.kw = B, Cin, H, W, Cout, k, k
    d.stride_h = d.stride_w = st
    d.pad_h = d.pad_w = pd
    d.dil_h = d.dil_w = 1
    d.draws = E
    ks = ctypes.c_int32(0)
    need = _lib.lib().bbb_conv2d_chwn_splitk_scratch(ctypes.byref(d), 1 if lrt else 0, ctypes.byref(ks))
    return ks.value, need


def test_plan_splits_small_launches_only():
    for name, B, Cin, H, W, Cout, k, st, pd, want in LAYERS:
        ks, need = _plan(B, Cin, H, W, Cout, k, st, pd, 1)
        assert ks == want and (need > 0) == (want > 1), (name, ks, need)
        ks10, need10 = _plan(B, Cin, H, W, Cout, k, st, pd, 10)
        assert ks10 == 1 and need10 == 0, name                     # the metric's 10-draw launches are never split
    assert _plan(512, 384, 2, 2, 256, 3, 1, 1, 1, lrt=True)[0] == 4


@pytest.mark.parametrize("layer", LAYERS[1:5], ids=lambda l: l[0])
def test_split_launch_vs_unsplit_vs_oracle(layer):
    from bbb_hip import ops
    name, B, Cin, H, W, Cout, k, st, pd, want = layer
    rs = np.random.default_rng(7)
    x = rs.standard_normal((1, B, Cin, H, W)).astype(np.float32)
    w = (rs.standard_normal((1, Cout, Cin, k, k)) * 0.05).astype(np.float32)
    b = rs.standard_normal((1, Cout)).astype(np.float32)
    xd = torch.from_numpy(x).cuda().permute(0, 2, 3, 4, 1).contiguous()
    wd, bd = torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda()
    ys = ops.conv2d_chwn_forward(xd, wd, bd, st, pd, 1, act="relu")
    ys2 = ops.conv2d_chwn_forward(xd, wd, bd, st, pd, 1, act="relu")
    assert torch.equal(ys, ys2)                                     # fixed combine order: bitwise run to run
    saved, ops.split_k = ops.split_k, False
    try:
        yu = ops.conv2d_chwn_forward(xd, wd, bd, st, pd, 1, act="relu")
    finally:
        ops.split_k = saved
    scale = float(yu.abs().max())
    assert 0 < float((ys - yu).abs().max()) <= TOL * scale            # split really ran (different rounding), inside the bound
    ref = np.maximum(O.conv2d(x[0], w[0], b[0], st, pd, 1), 0.0)
    np.testing.assert_allclose(ys[0].permute(3, 0, 1, 2).cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * max(1.0, scale))


def test_lrt_split_launch_moments_and_samples():
    from bbb_hip import ops
    B, Cin, H, W, Cout, k = 512, 384, 2, 2, 256, 3
    rs = np.random.default_rng(3)
    x = torch.from_numpy(rs.random((1, Cin, H, W, B)).astype(np.float32)).cuda()
    wmu = torch.from_numpy((rs.standard_normal((Cout, Cin, k, k)) * 0.05).astype(np.float32)).cuda()
    wvar = torch.from_numpy((rs.random((Cout, Cin, k, k)) * 1e-3).astype(np.float32)).cuda()
    bmu = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32) * 0.1).cuda()
    bvar = torch.from_numpy(rs.random(Cout).astype(np.float32) * 1e-3).cuda()
    run = lambda: ops.lrt_conv2d_chwn_forward(x, wmu, wvar, bmu, bvar, 7, 3, 2, 1, 1, 1, want_moments=True, act="softplus")
    y, am, av = run()
    y2, _, _ = run()
    assert torch.equal(y, y2)
    saved, ops.split_k = ops.split_k, False
    try:
        yu, amu, avu = run()
    finally:
        ops.split_k = saved
    for a, b_ in ((am, amu), (av, avu), (y, yu)):
        assert 0 < float((a - b_).abs().max()) <= TOL * float(b_.abs().max())


@pytest.mark.parametrize("lt,classes", [("bbb", 10), ("lrt", 100)])
def test_models_with_split_launches_stay_inside_the_bound(lt, classes):
    """Whole AlexNet, bs 512: the loop of single-draw forwards (split launches) against one batched 10-draw launch (unsplit),
    and a second pass through the same scratch buffer (launches of different sizes alternate in it)."""
    from bbb_hip import ensemble, ops, rng, zoo
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, classes, PRI, lt, "softplus").cuda()
    rng.assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    assert ops.split_k
    with torch.no_grad():
        batched = ensemble._mc_logits_chwn(net, x, 10, 7, 3)[0]
        for rep in range(2):
            loop = torch.cat([ensemble._mc_logits_chwn(net, x, 1, 7, 3 + j)[0] for j in range(10)])
            err = float((loop - batched).abs().max())
            assert 0 < err <= TOL * float(batched.abs().max()), (rep, err)
            if rep == 0:
                first = loop
        assert torch.equal(first, loop)


def _split_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        k = int(rs.choice([3, 5]))
        H, W = int(rs.randint(1, 6)), int(rs.randint(1, 6))
        p_, d_, s_ = int(rs.randint(0, 3)), int(rs.choice([1, 1, 2])), int(rs.choice([1, 1, 2]))
        if H + 2 * p_ < d_ * (k - 1) + 1 or W + 2 * p_ < d_ * (k - 1) + 1:
            continue
        Cin, Cout = int(rs.choice([32, 64, 130, 192])), int(rs.choice([10, 64, 65, 130]))
        B, E = int(rs.choice([64, 136, 264])), int(rs.choice([1, 2]))
        if _plan(B, Cin, H, W, Cout, k, s_, p_, E)[0] < 2:
            continue
        out.append((B, Cin, H, W, Cout, k, s_, p_, d_, E))
    return out


@pytest.mark.parametrize("c", _split_cases(10, 77), ids=lambda c: "x".join(map(str, c)))
def test_split_launches_random_geometry_vs_oracle(c):
    """Ragged channel / image tiles, strides, dilations and paddings through the SPLIT launch (the plan says so for each case)."""
    from bbb_hip import ops
    B, Cin, H, W, Cout, k, s_, p_, d_, E = c
    rs = np.random.default_rng(sum(c))
    x = rs.standard_normal((E, B, Cin, H, W)).astype(np.float32)
    w = (rs.standard_normal((E, Cout, Cin, k, k)) * 0.1).astype(np.float32)
    b = rs.standard_normal((E, Cout)).astype(np.float32)
    xd = torch.from_numpy(x).cuda().permute(0, 2, 3, 4, 1).contiguous()
    y = ops.conv2d_chwn_forward(xd, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), s_, p_, d_, act="softplus")
    saved, ops.split_k = ops.split_k, False
    try:
        yu = ops.conv2d_chwn_forward(xd, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), s_, p_, d_, act="softplus")
    finally:
        ops.split_k = saved
    assert float((y - yu).abs().max()) <= TOL * max(1.0, float(yu.abs().max()))
    for e in range(E):
        pre = O.conv2d(x[e], w[e], b[e], s_, p_, d_)
        mag = O.conv2d(np.abs(x[e]), np.abs(w[e]), np.abs(b[e]), s_, p_, d_)
        got = y[e].permute(3, 0, 1, 2).cpu().numpy()
        assert (np.abs(got - O.softplus_act(pre)) <= 2e-5 * mag + 2e-6).all()
