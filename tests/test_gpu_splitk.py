"""Split contraction (bbb_conv2d_chwn_splitk_fwd / bbb_lrt_conv2d_chwn_splitk_fwd, on by default through ops.split_k).  Since
round 4 it is a property of the LAYER: the plan depends on the layer's geometry only, an output tile's k range is cut into 2-4
ranges whose partial sums are added in range order -- by the last of several workgroups for small launches, inside one
workgroup for large ones, SAME BITS -- so a draw computed alone equals the same draw inside a 10-draw launch, a work unit, a
batch shard.  Stated bounds: split order vs the plain fmaf chain differ by at most 4e-6 of max|output| (partial sums of up to
1536 terms rounded separately; measured 2.3e-6 on the models below); the oracle bound of the plain kernel (2e-5) holds
unchanged; bitwise reproducible run to run; differently sized launches share one scratch buffer."""
import ctypes

import numpy as np
import pytest
import torch

import bbb_numpy as O

pytestmark = pytest.mark.gpu
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
TOL = 4e-6

# AlexNet / CIFAR layers at one draw, bs 512 (B, Cin, H, W, Cout, k, stride, pad) -> expected k ranges
LAYERS = [("conv1", 512, 3, 32, 32, 64, 11, 4, 5, 1), ("conv2", 512, 64, 4, 4, 192, 5, 1, 2, 1), ("conv3", 512, 192, 2, 2, 384, 3, 1, 1, 1),
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


def test_plan_is_a_function_of_the_layer_geometry_only():
    for name, B, Cin, H, W, Cout, k, st, pd, want in LAYERS:
        ks, need = _plan(B, Cin, H, W, Cout, k, st, pd, 1)
        assert ks == want and (need > 0) == (want > 1), (name, ks, need)
        for B2, E2 in ((512, 10), (128, 5), (64, 1), (4096, 25), (8, 3)):
            ks2, need2 = _plan(B2, Cin, H, W, Cout, k, st, pd, E2)
            assert ks2 == want, (name, B2, E2, ks2)                 # batch, draws, partitioning never change the summation order
        ks10, need10 = _plan(B, Cin, H, W, Cout, k, st, pd, 10)
        assert need10 == 0, name                                    # ... only which form runs: 10-draw launches need no scratch
    assert _plan(512, 384, 2, 2, 256, 3, 1, 1, 1, lrt=True)[0] == 4


@pytest.mark.parametrize("layer", LAYERS[3:5], ids=lambda l: l[0])
def test_cross_workgroup_and_in_workgroup_forms_agree_bitwise(layer):
    """The same layer through launches of every size: E = 1 (cross-workgroup, scratch), E = 10 (in-workgroup, 64-image tiles),
    E = 40 (in-workgroup, 128-image tiles), a 64-image batch slice: draw j is the same tensor everywhere."""
    from bbb_hip import ops
    name, B, Cin, H, W, Cout, k, st, pd, want = layer
    rs = np.random.default_rng(11)
    E = 40
    x = torch.from_numpy(rs.standard_normal((E, Cin, H, W, B)).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.standard_normal((E, Cout, Cin, k, k)) * 0.05).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.standard_normal((E, Cout)).astype(np.float32)).cuda()
    big = ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
    ten = ops.conv2d_chwn_forward(x[:10], w[:10], b[:10], st, pd, 1, act="softplus")
    assert torch.equal(ten, big[:10])
    for j in (0, 7, 39):
        one = ops.conv2d_chwn_forward(x[j:j + 1], w[j:j + 1], b[j:j + 1], st, pd, 1, act="softplus")
        assert torch.equal(one[0], big[j]), (name, j)
        sl = ops.conv2d_chwn_forward(x[j:j + 1, ..., 64:128].contiguous(), w[j:j + 1], b[j:j + 1], st, pd, 1, act="softplus")
        assert torch.equal(sl[0], big[j][..., 64:128]), (name, j)
    saved, ops.split_k = ops.split_k, False
    try:
        plain = ops.conv2d_chwn_forward(x[:10], w[:10], b[:10], st, pd, 1, act="softplus")
    finally:
        ops.split_k = saved
    assert 0 < float((plain - ten).abs().max()) <= TOL * float(plain.abs().max())


def test_lrt_cross_workgroup_and_in_workgroup_forms_agree_bitwise():
    from bbb_hip import ops
    B, Cin, H, W, Cout, k = 512, 384, 2, 2, 256, 3
    rs = np.random.default_rng(5)
    E = 6
    x = torch.from_numpy(rs.random((E, Cin, H, W, B)).astype(np.float32)).cuda()
    wmu = torch.from_numpy((rs.standard_normal((Cout, Cin, k, k)) * 0.05).astype(np.float32)).cuda()
    wvar = torch.from_numpy((rs.random((Cout, Cin, k, k)) * 1e-3).astype(np.float32)).cuda()
    bmu = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32) * 0.1).cuda()
    bvar = torch.from_numpy(rs.random(Cout).astype(np.float32) * 1e-3).cuda()
    big = ops.lrt_conv2d_chwn_forward(x, wmu, wvar, bmu, bvar, 7, 3, 2, 1, 1, 1, want_moments=True, act="softplus")      # 768 items
    for j in (0, 5):
        one = ops.lrt_conv2d_chwn_forward(x[j:j + 1], wmu, wvar, bmu, bvar, 7, 3 + j, 2, 1, 1, 1, want_moments=True, act="softplus")
        for a, b_ in zip(one, big):
            assert torch.equal(a[0], b_[j])


@pytest.mark.parametrize("layer", LAYERS[3:5], ids=lambda l: l[0])
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
def test_models_loop_equals_batched_and_stays_inside_the_bound(lt, classes):
    """Whole AlexNet, bs 512, SHIPPED defaults: the loop of single-draw forwards (cross-workgroup split on conv4 / conv5)
    equals one batched 10-draw launch (in-workgroup form) bit for bit, a second pass through the same scratch buffer included
    (launches of different sizes alternate in it); against the plain fmaf chain (ops.split_k = False) inside the stated bound."""
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
            assert torch.equal(loop, batched), (rep, float((loop - batched).abs().max()))
        saved, ops.split_k = ops.split_k, False
        try:
            plain = ensemble._mc_logits_chwn(net, x, 10, 7, 3)[0]
        finally:
            ops.split_k = saved
        err = float((plain - batched).abs().max())
        assert 0 < err <= TOL * float(plain.abs().max()), err


def _split_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        k = int(rs.choice([3, 5]))
        H, W = int(rs.randint(1, 4)), int(rs.randint(1, 4))
        p_, d_, s_ = int(rs.randint(0, 3)), int(rs.choice([1, 1, 2])), int(rs.choice([1, 1, 2]))
        if H + 2 * p_ < d_ * (k - 1) + 1 or W + 2 * p_ < d_ * (k - 1) + 1:
            continue
        Cin, Cout = int(rs.choice([64, 130, 192, 384])), int(rs.choice([10, 64, 65, 130]))
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
