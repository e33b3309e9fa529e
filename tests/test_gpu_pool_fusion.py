"""Pooling in the GEMM launch (bbb_conv_desc_t::pool, pconv_body.cuh POOL): a conv layer followed by
[activation ->] MaxPool2d(2, 2) as ONE launch -- a workgroup walks the four conv pixels of a pooling window, keeps the running
maximum of act(conv + bias) in accumulation registers and stores the pooled row.  The matrix work is exactly the unfused
launch's and max is exact, so the result must be BITWISE maxpool_chwn(conv2d_chwn_forward(...), 2, 2); the ensemble path takes
the fused launch when ops.pool_fusion_ok says the launch is large enough, and its results do not change."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(E, Cin, H, Cout, k, s, p, B, bias=True, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(E, Cin, H, H, B, device="cuda", generator=g)
    w = torch.randn(E, Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    b = torch.randn(E, Cout, device="cuda", generator=g) if bias else None
    return x, w, b, (s, p, 1)


@pytest.mark.parametrize("geom", [
    dict(E=3, Cin=3, H=32, Cout=64, k=11, s=4, p=5, B=128),      # AlexNet conv1: 8x8 -> 4x4, ragged tap sets at every border
    dict(E=2, Cin=64, H=4, Cout=192, k=5, s=1, p=2, B=256),      # AlexNet conv2: 4x4 -> 2x2
    dict(E=2, Cin=16, H=6, Cout=100, k=3, s=1, p=1, B=96),       # ragged channel tile and ragged 128-image tile
    dict(E=1, Cin=8, H=12, Cout=32, k=3, s=2, p=0, B=4),         # no padding, 5x5 conv map is odd -> see the error test; here 12 -> k3 s2 -> 5
    dict(E=2, Cin=5, H=9, Cout=70, k=2, s=1, p=0, B=132),        # 8x8 map, kernel 2
])
@pytest.mark.parametrize("act", [None, "relu", "softplus"])
def test_fused_pool_is_conv_then_pool_bit_for_bit(geom, act):
    from bbb_hip import ops, _lib
    x, w, b, g3 = _case(**geom)
    y = ops.conv2d_chwn_forward(x, w, b, *g3, act=act)
    if y.shape[2] % 2 or y.shape[3] % 2:
        with pytest.raises(_lib.BBBHipError):
            ops.conv2d_chwn_forward(x, w, b, *g3, act=act, pool=True)
        return
    want = ops.maxpool_chwn(y, 2, 2)
    got = ops.conv2d_chwn_forward(x, w, b, *g3, act=act, pool=True)
    assert got.shape == want.shape and torch.equal(got, want)
    if geom["E"] > 1:
        # without bias, and with one input shared by the draws
        got2 = ops.conv2d_chwn_forward(x[:1], w, None, *g3, act=act, pool=True)
        want2 = ops.maxpool_chwn(ops.conv2d_chwn_forward(x[:1], w, None, *g3, act=act), 2, 2)
        assert torch.equal(got2, want2)


def test_fused_pool_with_steps_per_launch_and_work_units():
    """The slab mappings of the descriptor (x_unit_div / x_unit_off: several steps per launch; work units) under pool = 1."""
    from bbb_hip import ops
    E, D = 6, 3
    x, w, b, g3 = _case(2, 3, 32, 64, 11, 4, 5, 128, seed=3)          # 2 batches, 3 draws each
    w6 = torch.randn(E, 64, 3, 11, 11, device="cuda") * 0.1
    b6 = torch.randn(E, 64, device="cuda")
    want = ops.maxpool_chwn(ops.conv2d_chwn_forward(x, w6, b6, *g3, act="softplus", x_div=D), 2, 2)
    got = ops.conv2d_chwn_forward(x, w6, b6, *g3, act="softplus", x_div=D, pool=True)
    assert torch.equal(got, want)
    w5, b5 = w6[:5].contiguous(), b6[:5].contiguous()                 # a share that starts at draw 1 of the first batch
    want = ops.maxpool_chwn(ops.conv2d_chwn_forward(x, w5, b5, *g3, act="relu", x_div=D, x_off=1), 2, 2)
    got = ops.conv2d_chwn_forward(x, w5, b5, *g3, act="relu", x_div=D, x_off=1, pool=True)
    assert torch.equal(got, want)
    # work units: 2 slices per draw, units 1..4 of 3 draws -> weight sets of draws 0..2
    xs = torch.randn(2, 3, 32, 32, 64, device="cuda")
    kw = dict(units=(2, 1), n_units=4, x_per_slice=True)
    w3, b3 = w6[:3].contiguous(), b6[:3].contiguous()
    want = ops.maxpool_chwn(ops.conv2d_chwn_forward(xs, w3, b3, *g3, act="softplus", **kw), 2, 2)
    got = ops.conv2d_chwn_forward(xs, w3, b3, *g3, act="softplus", pool=True, **kw)
    assert torch.equal(got, want)


def test_split_layers_and_other_kernels_refuse_the_pool():
    from bbb_hip import ops, _lib
    x, w, b, g3 = _case(1, 384, 2, 256, 3, 1, 1, 128)                 # AlexNet conv4: the layer's contraction is split
    assert not ops.pool_fusion_ok(tuple(x.shape), tuple(w.shape), *g3, 4000)
    with pytest.raises(_lib.BBBHipError):
        ops.conv2d_chwn_forward(x, w, b, *g3, pool=True)
    x, w, b, g3 = _case(2, 3, 32, 64, 11, 4, 5, 128)
    with pytest.raises(_lib.BBBHipError):
        ops.conv2d_chwn_forward(x, w, b, *g3, pool=True, bf16x3=True)


def test_ensemble_takes_the_fused_launch_for_large_launches_and_keeps_its_bits():
    """AlexNet bs 512 x 10 draws x 4 steps per launch: conv1 + softplus + pool1 is one launch (2560 items, 10 per CU); conv2's
    pooled form would keep eight 410 KB weight tiles live per XCD and stays separate; conv5's contraction is split.  One step
    per launch with nothing else in flight is too small (640 items) -- unless the caller says other lanes run beside it
    (ops.overlapped_launches).  Same logits."""
    import ref_port_torch as P
    from bbb_hip import ensemble, ops, rng, zoo
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    rng.assign_stream_ids(net)
    x = torch.rand(4 * 512, 3, 32, 32, device="cuda")
    saved = ops.pool_fusion
    try:
        res = {}
        for fuse in (True, False):
            ops.pool_fusion = fuse
            t = ensemble.Timers()
            with torch.no_grad():
                lg, kl = ensemble._mc_logits_chwn(net, x, 10, 5, 40, timers=t, groups=4)
                t1 = ensemble.Timers()
                one, _ = ensemble._mc_logits_chwn(net, x[:512], 10, 5, 40, timers=t1)
                t2 = ensemble.Timers()
                with ops.overlapped_launches():
                    lane, _ = ensemble._mc_logits_chwn(net, x[:512], 10, 5, 40, timers=t2)
            torch.cuda.synchronize()
            res[fuse] = (lg.clone(), t.summary()["maxpool"]["n"], one.clone(), t1.summary()["maxpool"]["n"], lane.clone(),
                         t2.summary()["maxpool"]["n"])
    finally:
        ops.pool_fusion = saved
    assert res[True][1] == 2 and res[False][1] == 3                   # pool1 went into conv1's launch
    assert res[True][3] == 3 and res[True][5] == 2                    # a lone 10-draw step keeps it; a lane of a pipeline does not
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][0][:10], res[True][2]) and torch.equal(res[True][2], res[True][4])   # ... all the same bits


@pytest.mark.parametrize("geom", [
    dict(E=3, Cin=3, H=32, Cout=64, k=11, s=4, p=5, B=128),      # AlexNet conv1 (LRT): 8x8 -> 4x4
    dict(E=2, Cin=64, H=4, Cout=192, k=5, s=1, p=2, B=256),      # AlexNet conv2 (LRT): 4x4 -> 2x2
    dict(E=2, Cin=16, H=6, Cout=100, k=3, s=1, p=1, B=96),       # ragged channel tile and ragged 64-image tile
])
@pytest.mark.parametrize("act", [None, "relu", "softplus"])
def test_fused_pool_lrt_is_lrt_conv_then_pool_bit_for_bit(geom, act):
    """pool = 1 on the LRT launch (round 5): per window pixel the sampling epilogue act(act_mu + sqrt(act_var) * eps) with the
    noise element of the CONV output's canonical NCHW index -- the Philox stream of the unpooled launch -- then the running maximum."""
    from bbb_hip import ops, _lib
    g = torch.Generator(device="cuda").manual_seed(7)
    E, Cin, H, Cout, k, B = geom["E"], geom["Cin"], geom["H"], geom["Cout"], geom["k"], geom["B"]
    x = torch.randn(E, Cin, H, H, B, device="cuda", generator=g)
    w_mu = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    w_var = torch.rand(Cout, Cin, k, k, device="cuda", generator=g) * 1e-2
    b_mu = torch.randn(Cout, device="cuda", generator=g)
    b_var = torch.rand(Cout, device="cuda", generator=g) * 1e-2
    g3 = (geom["s"], geom["p"], 1)
    for sample in (True, False):
        y, _, _ = ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 99, 5, 17, *g3, sample=sample, act=act, b_offset=3)
        want = ops.maxpool_chwn(y, 2, 2)
        got, _, _ = ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 99, 5, 17, *g3, sample=sample, act=act, b_offset=3, pool=True)
        assert got.shape == want.shape and torch.equal(got, want), (sample, float((got - want).abs().max()))
    # external noise (the replay entry), no bias
    eps = torch.randn(E, Cout, y.shape[2], y.shape[3], B, device="cuda", generator=g)
    y, _, _ = ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, None, None, 99, 5, 17, *g3, eps=eps, act=act)
    got, _, _ = ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, None, None, 99, 5, 17, *g3, eps=eps, act=act, pool=True)
    assert torch.equal(got, ops.maxpool_chwn(y, 2, 2))
    with pytest.raises(_lib.BBBHipError):
        ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 99, 5, 17, *g3, act=act, pool=True, want_moments=True)
