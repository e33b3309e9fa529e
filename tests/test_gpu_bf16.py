"""bf16 storage path (BASELINE.json configs[1]: "Bayesian3Conv3FC CIFAR-10, BBB layers, bf16, batch 256") on the MI355X
against the oracle's bf16 storage model and against the fp32 path.  Stated tolerances:
  * one GEMM launch, fp32 output, same bf16 operands: 2e-5 relative to sum_k |w||x| (fp32 accumulation order only);
  * one GEMM launch, bf16 output: half a bf16 ulp of the result (2^-9 relative) on top of that;
  * whole model vs the oracle's bf16 model (same rounding points, fp64 accumulate): a hidden activation that lands
    within accumulation error of a rounding boundary flips by one bf16 ulp (0.4 %) and the flips spread through the
    next layers -> logits within 1e-2 * max|logit| (measured <= 6e-3);
  * whole Monte-Carlo step vs the fp32 path (the reference's arithmetic): log-probabilities within 2e-2 of their
    largest magnitude (measured: 0.5 %), KL identical (it never touches bf16).
Run with -m gpu."""
import numpy as np
import pytest
import torch

import bbb_numpy as O
import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo)


def _bf(t):
    return t.to(torch.bfloat16)


def _pack_w(w, tap_major=False):
    """[E, Cout, Cin, kh, kw] fp32 -> the kernel's operand: bf16 [E, Cout, Kp], zero pad; tap-major = (r, q, ci) columns."""
    E, Cout = w.shape[:2]
    K = w[0, 0].numel()
    Kp = (K + 7) & ~7
    out = torch.zeros(E, Cout, Kp, dtype=torch.bfloat16, device=w.device)
    src = w.permute(0, 1, 3, 4, 2) if tap_major else w
    out[:, :, :K] = _bf(src.reshape(E, Cout, K))
    return out


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, E, x_shared
    (16, 3, 32, 32, 64, 11, 4, 5, 1, 2, True),      # AlexNet conv1: K = 363 (padded rows), one channel tile
    (24, 3, 16, 16, 32, 5, 1, 2, 1, 1, True),       # 3Conv3FC conv1: K = 75, B not a multiple of 128
    (136, 6, 9, 7, 70, 3, 1, 1, 1, 2, False),       # two batch tiles + ragged tile, Cout not a multiple of 64
    (8, 16, 6, 6, 130, 3, 2, 1, 2, 3, False),       # stride + dilation, three channel tiles
    (8, 1, 12, 12, 6, 5, 1, 0, 1, 1, True),         # LeNet conv1: K = 25
    (40, 520, 1, 1, 10, 1, 1, 0, 1, 2, False),      # linear layer, K = 520 (more than two 256-entry decode chunks)
    (8, 64, 4, 4, 64, 5, 1, 2, 1, 1, True),         # K = 1600: seven decode chunks
    (16, 24, 5, 7, 40, 3, 1, 1, 1, 2, False),       # cin = 24: 64-k tiles straddle taps in the tap-major order
    (264, 384, 2, 2, 256, 3, 1, 1, 1, 1, False),    # AlexNet conv4 shape: 4 of 9 taps in bounds
    (136, 1040, 1, 1, 10, 1, 1, 0, 1, 2, False),    # a classifier with a long row: a handful of workgroups, FOUR k-groups each
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,d,E,xs", CONV_CASES)
@pytest.mark.parametrize("out_f32,tap_major", [(True, False), (False, False), (False, True)])
def test_conv_bf16_vs_oracle(env, B, Cin, H, W, Cout, k, s, p, d, E, xs, out_f32, tap_major):
    if tap_major and Cin % 8 != 0:
        pytest.skip("tap-major rows need cin % 8 == 0")
    torch.manual_seed(B * 7 + Cout)
    x = torch.randn(1 if xs else E, B, Cin, H, W, device="cuda")
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * 0.2
    bias = torch.randn(E, Cout, device="cuda")
    xb = _bf(x.permute(0, 2, 3, 4, 1).contiguous())                       # [E|1, C, H, W, B]
    y = env["ops"].conv2d_chwn_bf16_forward(xb, _pack_w(w, tap_major), bias, (Cin, k, k), s, p, d, act="softplus", out_f32=out_f32,
                                            tap_major=tap_major)
    assert y.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    got = y.float().permute(0, 4, 1, 2, 3).cpu().numpy()                 # [E, B, Cout, Ho, Wo]
    xr = xb.float().permute(0, 4, 1, 2, 3).cpu().numpy()
    wr = _bf(w).float().cpu().numpy()
    for e in range(E):
        xe = xr[0 if xs else e]
        pre = O.conv2d(xe, wr[e], bias[e].cpu().numpy(), s, p, d)
        mag = O.conv2d(np.abs(xe), np.abs(wr[e]), np.abs(bias[e].cpu().numpy()), s, p, d)
        want = O.softplus_act(pre)
        tol = 2e-5 * mag + 2e-6                                          # softplus is 1-Lipschitz
        if not out_f32:
            tol = tol + np.abs(want) * 2.0 ** -8
        assert got[e].shape == want.shape
        err = np.abs(got[e] - want)
        assert (err <= tol).all(), f"draw {e}: max excess {(err - tol).max():.3e}"


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,E,xs", [
    (256, 3, 32, 32, 32, 5, 1, 2, 4, False),      # 3Conv3FC conv1, four one-draw steps in one launch: runs of 8 pixels
    (264, 3, 32, 32, 32, 5, 1, 2, 1, True),       # ragged second image tile, runs of 4
    (64, 1, 32, 32, 6, 5, 1, 0, 3, True),         # LeNet conv1: K = 25 (two MFMA steps), 6 channels
    (40, 6, 9, 7, 70, 3, 2, 1, 2, False),         # K = 54, three channel tiles of 32, stride 2, 63 % 16 != 0 pixels
    (16, 8, 10, 10, 40, 4, 1, 1, 2, False),       # K = 128: eight steps, two 32-channel register tiles
])
def test_short_row_first_layer_kernel_equals_the_general_kernel(env, B, Cin, H, W, Cout, k, s, p, E, xs):
    """Rows of <= 128 k with bf16 output run on pconv_bf16_smallk_kernel (weights in registers, a run of pixels per workgroup);
    the fp32-output form of the same launch runs on the general kernel.  Same MFMA sequence per element -> the rounded fp32
    output must equal the bf16 output bit for bit."""
    torch.manual_seed(B + Cout)
    x = _bf(torch.randn(1 if xs else E, Cin, H, W, B, device="cuda"))
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * 0.2
    bias = torch.randn(E, Cout, device="cuda")
    for act in ("softplus", "relu", None):
        y32 = env["ops"].conv2d_chwn_bf16_forward(x, _pack_w(w), bias, (Cin, k, k), s, p, 1, act=act, out_f32=True)
        y16 = env["ops"].conv2d_chwn_bf16_forward(x, _pack_w(w), bias, (Cin, k, k), s, p, 1, act=act, out_f32=False)
        assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape
        assert torch.equal(y32.to(torch.bfloat16), y16), float((y32 - y16.float()).abs().max())


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,E,xs,pk", [
    (256, 3, 32, 32, 32, 5, 1, 2, 16, False, 3),  # 3Conv3FC conv1 + MaxPool2d(3, 2), 16 one-draw steps per launch: strips of 5 pooled pixels
    (256, 3, 32, 32, 32, 5, 1, 2, 1, True, 3),    # one step: strips of 2 pooled pixels (the narrowest)
    (264, 3, 32, 32, 32, 5, 1, 2, 2, True, 3),    # ragged second image tile
    (64, 1, 32, 32, 6, 5, 1, 0, 3, True, 2),      # LeNet conv1 + MaxPool2d(2, 2): K = 25, 6 channels, 28 x 28 -> 14 x 14
    (40, 6, 9, 7, 70, 3, 1, 1, 2, False, 3),      # K = 54, three channel tiles, 9 x 7 -> 4 x 3 (odd widths: a conv column unused)
    (16, 8, 11, 10, 40, 4, 1, 1, 2, False, 2),    # K = 128: eight steps; 10 x 9 -> 5 x 4 (last conv row / column unused by 2 / 2)
    (8, 3, 8, 8, 32, 5, 1, 2, 1, True, 3),        # tiny map: 8 x 8 -> 3 x 3, one strip per row
])
def test_pooled_first_layer_equals_conv_then_pool(env, B, Cin, H, W, Cout, k, s, p, E, xs, pk):
    """bbb_conv_desc_t::pool on the bf16 path (pconv_bf16_smallk_pool_kernel): the maximum over the window's fp32 contraction
    results, then bias + activation + rounding once -- element for element what maxpool_chwn_bf16 makes of the unfused launch
    (x -> round(act(x + bias)) is non-decreasing)."""
    ops = env["ops"]
    torch.manual_seed(B + Cout + pk)
    x = _bf(torch.randn(1 if xs else E, Cin, H, W, B, device="cuda"))
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * 0.2
    bias = torch.randn(E, Cout, device="cuda")
    for act in ("softplus", "relu", None):
        y = ops.conv2d_chwn_bf16_forward(x, _pack_w(w), bias, (Cin, k, k), s, p, 1, act=act)
        want = ops.maxpool_chwn_bf16(y, pk, 2)
        got = ops.conv2d_chwn_bf16_forward(x, _pack_w(w), bias, (Cin, k, k), s, p, 1, act=act, pool=(pk, 2))
        assert got.dtype == torch.bfloat16 and got.shape == want.shape, (got.shape, want.shape)
        assert torch.equal(got, want), (act, float((got.float() - want.float()).abs().max()))
    # no bias
    y = ops.conv2d_chwn_bf16_forward(x, _pack_w(w), None, (Cin, k, k), s, p, 1, act="softplus")
    got = ops.conv2d_chwn_bf16_forward(x, _pack_w(w), None, (Cin, k, k), s, p, 1, act="softplus", pool=(pk, 2))
    assert torch.equal(got, ops.maxpool_chwn_bf16(y, pk, 2))


def test_pooled_first_layer_rejects_what_it_does_not_cover(env):
    ops = env["ops"]
    from bbb_hip import BBBHipError
    x = _bf(torch.randn(1, 64, 8, 8, 16, device="cuda"))
    w = torch.randn(1, 32, 64, 3, 3, device="cuda")
    with pytest.raises(BBBHipError):                                         # K = 576: not a short-row first layer
        ops.conv2d_chwn_bf16_forward(x, _pack_w(w), None, (64, 3, 3), 1, 1, 1, pool=(2, 2))
    x1 = _bf(torch.randn(1, 3, 16, 16, 16, device="cuda"))
    w1 = torch.randn(1, 8, 3, 3, 3, device="cuda")
    with pytest.raises(BBBHipError):                                         # 3 / 3 windows: not admitted
        ops.conv2d_chwn_bf16_forward(x1, _pack_w(w1), None, (3, 3, 3), 1, 1, 1, pool=(3, 3))
    import torch.nn as nn
    assert ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2)) and ops.bf16_pool_fusion_ok((1, 5, 5), False, False, nn.MaxPool2d(2))
    assert not ops.bf16_pool_fusion_ok((32, 5, 5), True, False, nn.MaxPool2d(3, 2))
    assert not ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2, padding=1))
    assert not ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2, ceil_mode=True))
    with ops.use_config(pool_fusion=False):
        assert not ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2))
    # the launch-size thresholds live in the LaunchConfig (0 = always: one of the library's two forms wins at every size)
    g3 = (1, 2, 1)
    assert ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2), (1, 3, 32, 32, 256), g3, 1)
    with ops.use_config(bf16_pool_fuse_min_rows=96, bf16_pool_fuse_min_rows_overlapped=60):
        assert ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2), (16, 3, 32, 32, 256), g3, 16)
        assert not ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2), (4, 3, 32, 32, 256), g3, 4)
        with ops.overlapped_launches():
            assert ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2), (4, 3, 32, 32, 256), g3, 4)
            assert not ops.bf16_pool_fusion_ok((3, 5, 5), False, False, nn.MaxPool2d(3, 2), (1, 3, 32, 32, 256), g3, 1)


@pytest.mark.parametrize("E,H,W,Cout,pad,B,xs", [
    (16, 15, 15, 64, 2, 256, False),     # 3Conv3FC conv2 at 16 steps per launch
    (3, 9, 13, 48, 2, 200, False),       # ragged image tile (200 = 128 + 72), a partial channel tile, a width that is no multiple of 3
    (2, 12, 16, 64, 0, 128, True),       # no padding, one input slab shared by the draws
    (5, 8, 11, 100, 1, 136, False),      # two channel tiles (the second ragged), padding 1
    (4, 5, 5, 64, 2, 256, False),        # a 5 x 5 map: every pixel is a border pixel
    (2, 9, 9, 64, 4, 32, False),         # padding 4: up to four of five taps outside on either side
    (1, 6, 4, 8, 2, 8, False),           # one strip narrower than its width, 8 images, 8 channels
])
def test_strip_form_over_channel_interleaved_input_equals_the_general_kernel(env, E, H, W, Cout, pad, B, xs):
    """BBB_BF16_X_C8 / pconv_bf16_strip8_kernel (3Conv3FC conv2's shape family: tap-major rows of 32 channels, 5 x 5, stride 1): the
    strip form reads [C / 8][H][W][B][8] activations straight into its MFMA operands and issues, per output element, the general
    kernel's one-k-group MFMA sequence -- so its result is that kernel's, bit for bit, in either output layout and WHATEVER the
    launch size (the general kernel's small launches differ from its large ones by rounding)."""
    ops = env["ops"]
    torch.manual_seed(E * 100 + W)
    x = _bf(torch.rand(1 if xs else E, 32, H, W, B, device="cuda"))
    w = _pack_w(torch.randn(E, Cout, 32, 5, 5, device="cuda") * 0.05, tap_major=True)
    bias = torch.randn(E, Cout, device="cuda") * 0.1
    x8 = ops.to_c8(x)
    assert x8.shape == (x.shape[0], 4, H, W, B, 8) and torch.equal(ops.from_c8(x8), x)
    # the reference launch: the general kernel with ONE k-group, i.e. as part of a launch of >= 512 workgroups (smaller launches
    # split a pixel's contraction over k-groups: another summation order) -- the same operands repeated along the draw dimension
    R = -(-1100 // (E * x.shape[2] * x.shape[3]))
    rep = lambda t: None if t is None else t.repeat(R, *([1] * (t.dim() - 1)))
    for act, b in (("softplus", bias), ("relu", bias), (None, None)):
        want = ops.conv2d_chwn_bf16_forward(x if xs else rep(x), rep(w), rep(b), (32, 5, 5), 1, pad, 1, act=act, tap_major=True)[:E]
        small = ops.conv2d_chwn_bf16_forward(x, w, b, (32, 5, 5), 1, pad, 1, act=act, tap_major=True)
        assert float((small.float() - want.float()).abs().max()) <= 2.0 ** -6 * max(1.0, float(want.float().abs().max()))
        got = ops.conv2d_chwn_bf16_forward(x8, w, b, (32, 5, 5), 1, pad, 1, act=act, tap_major=True)
        assert got.shape == want.shape and torch.equal(got, want), (act, float((got.float() - want.float()).abs().max()))
        if Cout % 8 == 0:
            got8 = ops.conv2d_chwn_bf16_forward(x8, w, b, (32, 5, 5), 1, pad, 1, act=act, tap_major=True, out_c8=True)
            assert got8.shape == (E, Cout // 8, want.shape[2], want.shape[3], B, 8)
            assert torch.equal(ops.from_c8(got8), want), act


def test_strip_form_work_units_and_grouped_steps(env):
    """The strip form under the launch shapes of the batched path: several steps per launch (output slab e reads input slab e) and
    a rank's work units (unit u = draw u // S with batch slice u % S) -- against the general kernel on the same descriptors."""
    ops = env["ops"]
    torch.manual_seed(5)
    S, E, B = 2, 3, 64                                  # 2 batch slices x 3 draws = 6 units; this rank holds units 1 .. 4
    xs = _bf(torch.rand(S, 32, 7, 9, B, device="cuda"))
    w = _pack_w(torch.randn(E, 64, 32, 5, 5, device="cuda") * 0.05, tap_major=True)
    bias = torch.randn(E, 64, device="cuda") * 0.1
    kw = dict(units=(S, 1), n_units=4, x_per_slice=True)     # (slices, first unit)
    got = ops.conv2d_chwn_bf16_forward(ops.to_c8(xs), w, bias, (32, 5, 5), 1, 2, 1, act="softplus", tap_major=True, **kw)
    # unit u of this rank = global unit 1 + u: draw (1 + u) // S, batch slice (1 + u) % S -- as plain launches of the strip form
    for u in range(4):
        e, sl = (1 + u) // S, (1 + u) % S
        one = ops.conv2d_chwn_bf16_forward(ops.to_c8(xs[sl:sl + 1]), w[e:e + 1], bias[e:e + 1], (32, 5, 5), 1, 2, 1, act="softplus", tap_major=True)
        assert torch.equal(got[u], one[0]), u
    want = ops.conv2d_chwn_bf16_forward(xs, w, bias, (32, 5, 5), 1, 2, 1, act="softplus", tap_major=True, **kw)
    assert float((got.float() - want.float()).abs().max()) <= 2.0 ** -6 * max(1.0, float(want.float().abs().max()))


@pytest.mark.parametrize("B,E,pk", [(256, 16, 3), (256, 2, 3), (64, 1, 2), (200, 5, 3)])
def test_pooled_first_layer_writes_the_channel_interleaved_layout(env, B, E, pk):
    """BBB_BF16_OUT_C8 on both pooled first-layer forms (the strip form of large launches, the window-resident form of small ones):
    the same values as the batch-innermost output, 8 channels of an image adjacent."""
    ops = env["ops"]
    torch.manual_seed(B + E)
    x = _bf(torch.rand(E, 3, 32, 32, B, device="cuda"))
    w = _pack_w(torch.randn(E, 32, 3, 5, 5, device="cuda") * 0.2)
    bias = torch.randn(E, 32, device="cuda")
    for act in ("softplus", None):
        want = ops.conv2d_chwn_bf16_forward(x, w, bias, (3, 5, 5), 1, 2, 1, act=act, pool=(pk, 2))
        got = ops.conv2d_chwn_bf16_forward(x, w, bias, (3, 5, 5), 1, 2, 1, act=act, pool=(pk, 2), out_c8=True)
        assert got.shape == (E, 4, want.shape[2], want.shape[3], B, 8)
        assert torch.equal(ops.from_c8(got), want), act


def test_channel_interleaved_forms_reject_what_they_do_not_cover(env):
    ops = env["ops"]
    from bbb_hip import BBBHipError
    x8 = ops.to_c8(_bf(torch.rand(1, 64, 6, 6, 16, device="cuda")))
    w = _pack_w(torch.randn(1, 64, 64, 5, 5, device="cuda"), tap_major=True)
    with pytest.raises(BBBHipError):                                         # 64 input channels: no strip form
        ops.conv2d_chwn_bf16_forward(x8, w, None, (64, 5, 5), 1, 2, 1, tap_major=True)
    x = _bf(torch.rand(1, 32, 6, 6, 16, device="cuda"))
    w32 = _pack_w(torch.randn(1, 64, 32, 5, 5, device="cuda"), tap_major=True)
    with pytest.raises(BBBHipError):                                         # the general kernel writes batch-innermost only
        ops.conv2d_chwn_bf16_forward(x, w32, None, (32, 5, 5), 1, 2, 1, tap_major=True, out_c8=True)
    with pytest.raises(BBBHipError):                                         # stride 2
        ops.conv2d_chwn_bf16_forward(ops.to_c8(x), w32, None, (32, 5, 5), 2, 2, 1, tap_major=True)
    assert ops.bf16_c8_input_ok((32, 5, 5), (1, 2, 1), True, False)
    assert not ops.bf16_c8_input_ok((32, 5, 5), (1, 2, 1), False, False)     # reference-order rows
    assert not ops.bf16_c8_input_ok((32, 5, 5), (1, 2, 1), True, True)       # fp32 logits
    assert not ops.bf16_c8_input_ok((64, 5, 5), (1, 1, 1), True, False) and not ops.bf16_c8_input_ok((32, 3, 3), (1, 1, 1), True, False)
    assert not ops.bf16_c8_input_ok((32, 5, 5), (2, 2, 1), True, False) and not ops.bf16_c8_input_ok((32, 5, 5), (1, 5, 1), True, False)
    with ops.use_config(bf16_c8=False):
        assert not ops.bf16_c8_input_ok((32, 5, 5), (1, 2, 1), True, False)


@pytest.mark.parametrize("B,E,G", [(256, 1, 1), (256, 4, 1), (256, 1, 4), (256, 1, 16)])
def test_3conv3fc_bf16_with_and_without_the_interleaved_layout(env, B, E, G):
    """Whole model: conv1 + pool1 hand their output to conv2 channel-interleaved (LaunchConfig.bf16_c8) -- the logits are the same
    bits as on the batch-innermost layout throughout (where the general kernel runs conv2 with one k-group), eager, captured, and
    with several steps per launch."""
    ops, ens = env["ops"], env["ens"]
    torch.manual_seed(3)
    net = env["zoo"].BBB3Conv3FC(10, 3, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B * G, 3, 32, 32, device="cuda")
    outs = {}
    for on in (False, True):
        with torch.no_grad(), ops.use_config(bf16_c8=on):
            env["rng"].manual_seed(11, call=0)
            if G == 1:
                lo, kl = ens.mc_forward(net, x, E, precision="bf16")
            else:
                pipe = ens.GraphedPipeline(net, x[:B], E, depth=1, steps_per_launch=G, precision="bf16")
                bufs = [pipe.step(x[g * B:(g + 1) * B])[0] for g in range(G)]
                pipe.sync()
                lo = torch.stack([b.clone() for b in bufs])
            outs[on] = lo.clone()
    if E * G >= 4:      # conv2 on the general kernel: >= 512 workgroups, one k-group -- the order the strip form always uses
        assert torch.equal(outs[True], outs[False])
    else:               # the general kernel's small launches split a pixel's contraction over k-groups: rounding-level differences
        scale = max(1.0, float(outs[False].abs().max()))
        assert float((outs[True] - outs[False]).abs().max()) <= 2e-2 * scale


@pytest.mark.parametrize("E,B,K,Cout,out_f32,act", [
    (16, 256, 1000, 10, True, None),          # 3Conv3FC fc3 at 16 steps per launch
    (3, 40, 520, 10, True, "softplus"),       # ragged image tile, a row that is no multiple of the slice length
    (2, 136, 1040, 16, False, "relu"),        # 16 outputs, bf16 output
    (1, 8, 512, 7, True, None),               # the smallest row that takes this kernel, one image group
])
def test_few_output_classifier_kernel(env, E, B, K, Cout, out_f32, act):
    """pconv_bf16_fewout_kernel (<= 16 outputs, rows of >= 512: 3Conv3FC fc3): fp32 FMAs over k slices added in a fixed order --
    against the exact contraction of the same bf16 operands, and the same bits whether a draw is launched alone or with others."""
    ops = env["ops"]
    torch.manual_seed(K + Cout)
    x = _bf(torch.randn(E, K, 1, 1, B, device="cuda"))
    w = torch.randn(E, Cout, K, 1, 1, device="cuda") / K ** 0.5
    bias = torch.randn(E, Cout, device="cuda")
    wp = _pack_w(w)
    got = ops.conv2d_chwn_bf16_forward(x, wp, bias, (K, 1, 1), 1, 0, 1, act=act, out_f32=out_f32)
    assert got.shape == (E, Cout, 1, 1, B) and got.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    ref = torch.einsum("enk,ekb->enb", wp[:, :, :K].double(), x[:, :, 0, 0].double()) + bias[:, :, None].double()
    ref = {None: lambda t: t, "relu": torch.relu, "softplus": torch.nn.functional.softplus}[act](ref)
    tol = (2e-5 if out_f32 else 2.0 ** -8) * max(1.0, float(ref.abs().max()))
    assert float((got.double()[:, :, 0, 0] - ref).abs().max()) <= tol
    one = ops.conv2d_chwn_bf16_forward(x[E - 1:], wp[E - 1:], bias[E - 1:], (K, 1, 1), 1, 0, 1, act=act, out_f32=out_f32)
    assert torch.equal(one[0], got[E - 1])
    nob = ops.conv2d_chwn_bf16_forward(x, wp, None, (K, 1, 1), 1, 0, 1, act=None, out_f32=out_f32)
    ref0 = torch.einsum("enk,ekb->enb", wp[:, :, :K].double(), x[:, :, 0, 0].double())
    assert float((nob.double()[:, :, 0, 0] - ref0).abs().max()) <= tol


def test_sampled_weights_bf16_are_the_rounded_fp32_samples(env):
    """Same Philox stream, same fp32 arithmetic, one nearest-even rounding; pad columns untouched (zero); biases fp32."""
    torch.manual_seed(0)
    shapes = [(64, 3, 11, 11), (64,), (10, 33), (10,), (16, 8, 3, 3), (16,), (7, 5), (7,)]
    mus = [torch.randn(s, device="cuda") * 0.1 for s in shapes]
    rhos = [torch.randn(s, device="cuda") * 0.3 - 4 for s in shapes]
    ids = list(range(len(shapes)))
    E = 3
    ws, _, kl32 = env["ops"].reparam_kl_forward(mus, rhos, 0.0, 0.1, ids, 99, 5, draws=E)
    kl16, outs = env["ops"].sample_weights_bf16(mus, rhos, 0.0, 0.1, ids, 99, 5, E)
    assert torch.equal(kl16, kl32)
    for w32, o, s in zip(ws, outs, shapes):
        if len(s) == 1:
            assert o.dtype == torch.float32 and torch.equal(o, w32)
            continue
        K = int(np.prod(s[1:]))
        assert o.dtype == torch.bfloat16 and o.shape == (E, s[0], (K + 7) & ~7)
        src = w32.permute(0, 1, 3, 4, 2) if env["ops"].bf16_tap_major(s) else w32      # (r, q, ci) columns when cin % 8 == 0
        assert torch.equal(o[:, :, :K], _bf(src.reshape(E, s[0], K)))
        assert not o[:, :, K:].any()


def test_layout_and_pool_bf16_exact(env):
    torch.manual_seed(1)
    x = torch.randn(24, 5, 9, 11, device="cuda")
    xb = env["ops"].to_batch_innermost_bf16(x)
    assert torch.equal(xb, _bf(x).permute(1, 2, 3, 0).contiguous())
    for k, s in ((2, 2), (3, 2), (3, 1)):
        got = env["ops"].maxpool_chwn_bf16(xb.unsqueeze(0), k, s)[0]
        want = torch.nn.functional.max_pool2d(_bf(x).float(), k, s).to(torch.bfloat16).permute(1, 2, 3, 0)
        assert torch.equal(got, want.contiguous())


def _oracle_eps_fn(names, seed, call):
    idx = {n: i for i, n in enumerate(names)}

    def fn(name, kind, shape):
        stream = 4 * idx[name] + {"W": 0, "bias": 1}[kind]
        return O.normal_eps(seed, call, stream, int(np.prod(shape))).reshape(shape)
    return fn


@pytest.mark.parametrize("net_type,B,cin,hw", [("3conv3fc", 256, 3, 32), ("alexnet", 64, 3, 32), ("lenet", 32, 1, 32)])
def test_model_bf16_vs_oracle_bf16_model(env, net_type, B, cin, hw):
    """configs[1] shape (3Conv3FC, batch 256) + the other two topologies: every draw of the bf16 path vs the oracle's
    bf16 storage model under the same Philox noise."""
    torch.manual_seed(11)
    params = P.init_params(net_type, cin, 10, P.CONFIG_PRIORS)
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, "bbb", "softplus")
    sd = {f"{n}.{k}": v for n, p in params.items() if not n.startswith("_") for k, v in p.items()}
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, hw, hw)
    E, seed, call0 = 2, 4242, 7
    with torch.no_grad():
        logits, kl = env["ens"].mc_logits(net, x.cuda(), E, seed, call0, precision="bf16")
        logits32, kl32 = env["ens"].mc_logits(net, x.cuda(), E, seed, call0)
    assert torch.equal(kl, kl32)
    names = [op[1] for op in O.TOPOLOGY[net_type] if op[0] in ("conv", "fc")]
    npar = {n: {k: v.numpy() for k, v in p.items()} for n, p in params.items() if not n.startswith("_")}
    npar["_prior_mu"], npar["_prior_sigma"] = params["_prior_mu"], params["_prior_sigma"]
    for e in range(E):
        want, kl_o = O.model_forward_bf16(net_type, npar, x.numpy(), "softplus", _oracle_eps_fn(names, seed, call0 + e))
        got = logits[e].cpu().numpy()
        scale = float(np.abs(want).max())
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-2 * scale)
        assert abs(kl.item() - kl_o) <= 2e-6 * kl_o
        # and the bf16 result is a perturbation of the fp32 one, not something else
        np.testing.assert_allclose(got, logits32[e].cpu().numpy(), rtol=0, atol=6e-2 * scale)


def test_mc_step_bf16_close_to_fp32_and_graph_replays(env):
    """The whole step (sample -> layers -> log_softmax -> logmeanexp) in bf16 vs fp32 with the same noise; the captured
    graph reproduces the eager bf16 step bit for bit and draws fresh noise per replay."""
    torch.manual_seed(2)
    net = env["zoo"].BBBAlexNet(10, 3, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(64, 3, 32, 32, device="cuda")
    E = 10
    with torch.no_grad():
        env["rng"].manual_seed(7, call=0)
        lo32, kl32 = env["ens"].mc_forward(net, x, E)
        env["rng"].manual_seed(7, call=0)
        lo16, kl16 = env["ens"].mc_forward(net, x, E, precision="bf16")
        lo16b, _ = env["ens"].mc_forward(net, x, E, precision="bf16")        # next step: calls E .. 2E-1
    assert torch.equal(kl16, kl32)
    scale = max(1.0, float(lo32.abs().max()))           # random-init logits are O(100): compare on that scale
    assert float((lo16 - lo32).abs().max()) <= 2e-2 * scale
    assert float(lo16.exp().sum(1).sub(1).abs().max()) < 1e-4 * E            # still a mixture of distributions
    env["rng"].manual_seed(7, call=0)
    g = env["ens"].GraphedMC(net, x, E, precision="bf16")
    a, _ = g.step()
    a = a.clone()
    b, _ = g.step()
    torch.cuda.synchronize()
    assert torch.equal(a, lo16) and torch.equal(b, lo16b)


def test_bf16_refuses_what_it_does_not_cover(env):
    from bbb_hip._lib import BBBHipError
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "lrt", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    with torch.no_grad(), pytest.raises(BBBHipError):
        env["ens"].mc_forward(net, torch.rand(8, 1, 32, 32, device="cuda"), 2, precision="bf16")      # LRT layers
    net = env["zoo"].BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    with torch.no_grad(), pytest.raises(BBBHipError):
        env["ens"].mc_forward(net, torch.rand(12, 1, 32, 32, device="cuda"), 2, precision="bf16")     # B % 8 != 0
    with pytest.raises(BBBHipError):
        env["ens"].mc_forward(net, torch.rand(8, 1, 32, 32, device="cuda"), 2, precision="bf16")      # autograd on


@pytest.mark.parametrize("net_type,cin,B", [("3conv3fc", 3, 64), ("alexnet", 3, 32), ("lenet", 1, 16)])
def test_dropin_loop_in_bf16(env, net_type, cin, B):
    """ops.LaunchConfig.dropin_precision = "bf16": the unmodified `for j in range(num_ens): net(x)` loop (main_bayesian.py:73-80)
    runs the bf16 storage path -- call j returns draw j of the batched bf16 launch under the same noise calls, KL as in fp32."""
    ops, ens, rng = env["ops"], env["ens"], env["rng"]
    torch.manual_seed(3)
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    rng.assign_stream_ids(net)
    x = torch.rand(B, cin, 32, 32, device="cuda")
    E = 5
    with torch.no_grad():
        rng.manual_seed(11, call=4)
        want, kl_want = ens._mc_logits_chwn(net, x, E, 11, 4, precision="bf16")          # [E, C, B]
        rng.manual_seed(11, call=4)
        with ops.use_config(dropin_precision="bf16"):
            outs = [net(x) for _ in range(E)]
        rng.manual_seed(11, call=4)
        fp32 = [net(x)[0] for _ in range(E)]
    for j, (o, kl) in enumerate(outs):
        assert o.dtype == torch.float32 and o.shape == (B, 10)
        assert torch.equal(o, want[j].t()), j
        assert torch.equal(kl.reshape(()), kl_want.reshape(()))
        # bf16 storage of weights and activations: close to, and different from, the fp32 forward of the same draw
        d = float((o - fp32[j]).abs().max())
        assert 0.0 < d <= 3e-2 * float(fp32[j].abs().max()) + 1e-3, (j, d)
    rng.use_device_generator()


def test_dropin_bf16_refuses_what_it_does_not_cover(env):
    from bbb_hip._lib import BBBHipError
    ops = env["ops"]
    torch.manual_seed(4)
    net = env["zoo"].getModel("3conv3fc", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    lrt = env["zoo"].getModel("3conv3fc", 3, 10, P.CONFIG_PRIORS, "lrt", "softplus").cuda()
    x = torch.rand(16, 3, 32, 32, device="cuda")
    with ops.use_config(dropin_precision="bf16"):
        with pytest.raises(BBBHipError):
            net(x)                                                   # autograd enabled: bf16 is inference only
        with torch.no_grad():
            with pytest.raises(BBBHipError):
                lrt(x)                                               # LRT layers
            with pytest.raises(BBBHipError):
                net(x[:12])                                          # B % 8 != 0
            with pytest.raises(BBBHipError):
                net.conv1(x)                                         # a layer on its own
            h = net.conv2.register_forward_hook(lambda m, i, o: None)
            try:
                with pytest.raises(BBBHipError):
                    net(x)                                           # hooks: the layer-by-layer path, fp32 only
            finally:
                h.remove()
            out, kl = net(x)                                         # and what it does cover still runs
            assert out.shape == (16, 10) and torch.isfinite(out).all()
    with torch.no_grad():
        out32, _ = net(x)                                            # back in fp32 outside the context
    assert out32.dtype == torch.float32
