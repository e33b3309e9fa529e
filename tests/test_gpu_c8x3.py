"""Split-bf16 contraction over MFMA-ready operands (bbb_conv2d_c8x3_fwd, csrc/pconv_c8x3.hip): channel-interleaved split
activations ("c8 S3") + tap-major fp32 weights.  Same arithmetic as bbb_conv2d_chwn_bf16x3_fwd, held to the SAME bound as the fp32
kernel against the float64 oracle (4e-6 of sum_k |w||x|) on every operand scale.  Run with -m gpu."""
import numpy as np
import pytest
import torch

import bbb_numpy as O

pytestmark = pytest.mark.gpu
TOL = 4e-6


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo)


def test_c8s3_layout_round_trip_and_definition(env):
    """c8 S3 is a storage format of the same fp32 values: [E, 3, C/8, H, W, B, 8] holds bf16(a), bf16(a - hi), bf16(a - hi - mid)
    of channel 8g + i of image b at [e, :, g, h, w, b, i]; the round trip is exact on every scale."""
    ops = env["ops"]
    torch.manual_seed(0)
    x = (torch.randn(2, 24, 3, 5, 12, device="cuda") * torch.logspace(-12, 12, 24, device="cuda").view(1, 24, 1, 1, 1)).contiguous()
    c = ops.c8s3_from_f32(x)
    assert c.shape == (2, 3, 3, 3, 5, 12, 8) and c.dtype == torch.bfloat16
    assert torch.equal(ops.c8s3_to_f32(c), x)
    xr = x.view(2, 3, 8, 3, 5, 12).permute(0, 1, 3, 4, 5, 2)                 # [E, G, H, W, B, 8]
    hi = xr.to(torch.bfloat16)
    mid = (xr - hi.float()).to(torch.bfloat16)
    lo = (xr - hi.float() - mid.float()).to(torch.bfloat16)
    assert torch.equal(c[:, 0], hi) and torch.equal(c[:, 1], mid) and torch.equal(c[:, 2], lo)
    # pooling acts element-wise on the 16-byte channel vectors
    p = ops.c8s3_to_f32(ops.maxpool_c8s3(ops.c8s3_from_f32(x[:, :, :3, :4]), 2, 1))
    assert torch.equal(p, ops.maxpool_chwn(x[:, :, :3, :4].contiguous(), 2, 1))


def test_tap_major_weights(env):
    ops = env["ops"]
    w = torch.randn(3, 5, 8, 3, 2, device="cuda")
    assert torch.equal(ops.w_tap_major(w), w.permute(0, 1, 3, 4, 2).reshape(3, 5, 6, 8).contiguous())


CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, E, x_shared
    (512, 64, 4, 4, 192, 5, 1, 2, 1, 2, False),      # AlexNet conv2: most taps of border pixels out of bounds; 256-image tiles
    (256, 192, 2, 2, 384, 3, 1, 1, 1, 2, False),     # AlexNet conv3
    (132, 32, 9, 7, 72, 3, 1, 1, 1, 2, False),       # ragged image tile, ragged channel tile
    (8, 64, 6, 6, 136, 3, 2, 1, 2, 3, False),        # stride + dilation
    (40, 512, 1, 1, 16, 1, 1, 0, 1, 2, False),       # linear, K = 512
    (64, 256, 2, 2, 256, 3, 1, 1, 1, 1, True),       # AlexNet conv4 shape, one draw
    (300, 32, 5, 5, 8, 5, 1, 0, 1, 2, True),         # input shared by the draws, a single output pixel, 8 output channels
    (260, 48, 10, 10, 64, 3, 1, 0, 1, 2, True),      # 48 channels (one and a half k tiles per tap: tiles straddle taps; odd step count)
    (36, 16, 5, 4, 40, 3, 1, 1, 1, 2, False),        # 16 channels: two taps per k tile, ragged in-bounds rectangles
    (64, 80, 3, 3, 96, 2, 1, 1, 1, 1, False),        # 80 channels, 2 x 2 taps, a 96-channel workgroup tile
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,d,E,xs", CASES)
@pytest.mark.parametrize("xscale,wscale", [(3.0, 0.2), (0.004, 0.0003), (1e-6, 1e-9), (1e6, 1e-12)])
@pytest.mark.parametrize("out_f32", [False, True])
def test_c8x3_launch_vs_oracle_and_fp32_kernel(env, B, Cin, H, W, Cout, k, s, p, d, E, xs, xscale, wscale, out_f32):
    ops = env["ops"]
    if out_f32 and xscale != 3.0:
        pytest.skip("the fp32 output form shares everything but the store: one operand scale")
    torch.manual_seed(B + Cout)
    x = torch.randn(1 if xs else E, Cin, H, W, B, device="cuda") * xscale
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * wscale
    bias = torch.randn(E, Cout, device="cuda") * (xscale * wscale)
    y = ops.conv2d_c8x3_forward(ops.c8s3_from_f32(x), ops.w_tap_major(w), bias, k, s, p, d, act=None, out_f32=out_f32)
    if not out_f32:
        y = ops.c8s3_to_f32(y)
    y32 = ops.conv2d_chwn_forward(x, w, bias, s, p, d, act=None, bf16x3=False)
    assert y.shape == y32.shape
    worst = worst32 = 0.0
    for e in range(E):
        xe = x[0 if xs else e].permute(3, 0, 1, 2).double().cpu().numpy()            # [B, C, H, W]
        we, be = w[e].double().cpu().numpy(), bias[e].double().cpu().numpy()
        want = O.conv2d(xe, we, be, s, p, d)
        mag = O.conv2d(np.abs(xe), np.abs(we), np.abs(be), s, p, d)
        got = y[e].permute(3, 0, 1, 2).double().cpu().numpy()
        got32 = y32[e].permute(3, 0, 1, 2).double().cpu().numpy()
        worst = max(worst, float((np.abs(got - want) / mag).max()))
        worst32 = max(worst32, float((np.abs(got32 - want) / mag).max()))
    print(f"relative to sum|w||x|: c8x3 {worst:.2e}, fp32 kernel {worst32:.2e}")
    assert worst <= TOL, (worst, worst32)
    assert worst <= max(4e-7, 3.0 * worst32)


@pytest.mark.parametrize("act", ["relu", "softplus"])
def test_c8x3_fused_activation_and_tile_shapes(env, act, monkeypatch):
    """Bias + activation in the epilogue = the fp32 kernel's epilogue on the contraction result; the 128- and 256-image workgroup
    tiles run the same MFMA sequence per output element: bit-identical outputs."""
    import os
    ops = env["ops"]
    torch.manual_seed(5)
    x = torch.randn(2, 64, 4, 4, 512, device="cuda")
    w = torch.randn(2, 72, 64, 3, 3, device="cuda") * 0.05
    bias = torch.randn(2, 72, device="cuda")
    xc, wt = ops.c8s3_from_f32(x), ops.w_tap_major(w)
    y = ops.c8s3_to_f32(ops.conv2d_c8x3_forward(xc, wt, bias, 3, 1, 1, 1, act=act))
    y32 = ops.conv2d_chwn_forward(x, w, bias, 1, 1, 1, act=act, bf16x3=False)
    assert float((y - y32).abs().max()) <= 2e-5 * float(y32.abs().max())
    pre = ops.c8s3_to_f32(ops.conv2d_c8x3_forward(xc, wt, bias, 3, 1, 1, 1, act=None))
    ref = torch.relu(pre) if act == "relu" else torch.nn.functional.softplus(pre)
    assert float((y - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,d,E", [(512, 64, 4, 4, 192, 5, 1, 2, 1, 3), (132, 48, 6, 6, 200, 3, 1, 0, 1, 2),
                                                      (256, 256, 2, 2, 128, 3, 1, 1, 1, 2), (64, 32, 3, 5, 16, 1, 1, 0, 1, 2)])
def test_c8x3_workgroup_tiles_run_the_same_contraction(env, B, Cin, H, W, Cout, k, s, p, d, E):
    """32 * NT channels x 128 / 256 images per workgroup (NT = 2, 3, 4): every output element sees the same sequence of matrix
    instructions -- all six shapes, and the shape the library picks, produce the same bits, c8 S3 and fp32 output alike."""
    ops = env["ops"]
    torch.manual_seed(Cout)
    x = ops.c8s3_from_f32(torch.randn(E, Cin, H, W, B, device="cuda"))
    w = ops.w_tap_major(torch.randn(E, Cout, Cin, k, k, device="cuda") * 0.05)
    bias = torch.randn(E, Cout, device="cuda")
    for of32 in (False, True):
        ref = ops.conv2d_c8x3_forward(x, w, bias, k, s, p, d, act="softplus", out_f32=of32)
        for nt in (2, 3, 4):
            for tile in (128, 256):
                got = ops.conv2d_c8x3_forward(x, w, bias, k, s, p, d, act="softplus", out_f32=of32, nt=nt, tile=tile)
                assert torch.equal(got, ref), (of32, nt, tile)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,d,E,xs", [(512, 48, 10, 10, 64, 3, 1, 1, 3, True), (72, 32, 9, 13, 40, 2, 1, 2, 2, False),
                                                        (256, 64, 8, 8, 96, 1, 2, 1, 2, False), (40, 16, 6, 6, 136, 3, 1, 1, 1, False)])
@pytest.mark.parametrize("act", [None, "softplus", "relu"])
def test_c8x3_pooled_launch_is_the_launch_then_the_pooling(env, B, Cin, H, W, Cout, k, s, d, E, xs, act):
    """BBB_C8X3_POOL ("parallel window": the four waves of a workgroup own the four pixels of a 2 x 2 window): bit for bit
    maxpool_c8s3(conv2d_c8x3_forward(...), 2, 2), for every workgroup tile shape; odd maps drop their last row / column as
    MaxPool2d does -- here: even maps only (the odd ones raise)."""
    ops = env["ops"]
    torch.manual_seed(Cin + Cout)
    x = ops.c8s3_from_f32(torch.randn(1 if xs else E, Cin, H, W, B, device="cuda"))
    w = ops.w_tap_major(torch.randn(E, Cout, Cin, k, k, device="cuda") * 0.1)
    bias = torch.randn(E, Cout, device="cuda")
    full = ops.conv2d_c8x3_forward(x, w, bias, k, s, 0, d, act=act)
    if full.shape[3] % 2 or full.shape[4] % 2:
        with pytest.raises(__import__("bbb_hip")._lib.BBBHipError):
            ops.conv2d_c8x3_forward(x, w, bias, k, s, 0, d, act=act, pool=True)
        return
    want = ops.maxpool_c8s3(full, 2, 2)
    got = ops.conv2d_c8x3_forward(x, w, bias, k, s, 0, d, act=act, pool=True)
    assert got.shape == want.shape and torch.equal(got, want)
    for nt in (2, 3, 4):
        for tile in (32, 64):
            assert torch.equal(ops.conv2d_c8x3_forward(x, w, bias, k, s, 0, d, act=act, pool=True, nt=nt, tile=tile), want), (nt, tile)


@pytest.mark.parametrize("N,C,H,W,Cout,k,s,p,blocks", [(64, 3, 32, 32, 64, 11, 4, 5, 1), (24, 3, 32, 32, 16, 11, 4, 5, 3),
                                                       (16, 1, 17, 21, 8, 5, 2, 1, 2), (8, 5, 12, 12, 24, 3, 3, 0, 1),
                                                       (20, 3, 20, 70, 16, 11, 4, 5, 1)])      # two column tiles, ragged image tile
def test_space_to_depth_operands_and_layer(env, N, C, H, W, Cout, k, s, p, blocks):
    """bbb_s2d_c8s3 / bbb_w_s2d_tap_major against their definition (include/bbb_hip.h) built with torch indexing, and the strided
    layer computed as the m x m stride-1 layer on them against the float64 oracle (bound of the fp32 kernel) -- plain and with the
    activation and the 2 x 2 pooling inside the launch."""
    ops = env["ops"]
    torch.manual_seed(N + k)
    x = torch.randn(N, C, H, W, device="cuda")
    w = torch.randn(2, Cout, C, k, k, device="cuda") * 0.1
    bias = torch.randn(2, Cout, device="cuda")
    m, cp, hb, wb, ho, wo = ops.s2d_geometry(C, k, s, p, 1, H, W)
    xs = ops.s2d_c8s3(x, blocks, k, s, p)
    assert xs.shape == (blocks, 3, cp // 8, hb, wb, N // blocks, 8)
    xf = ops.c8s3_to_f32(xs)                                                     # [blocks, C', Hb, Wb, Bs]
    xpad = torch.zeros(N, C, s * hb + s, s * wb + s, device="cuda")
    xpad[:, :, p:p + H, p:p + W] = x
    want = torch.zeros(N, cp, hb, wb, device="cuda")
    for c in range(C):
        for dy in range(s):
            for dx in range(s):
                want[:, (c * s + dy) * s + dx] = xpad[:, c, dy:dy + s * hb:s, dx:dx + s * wb:s]
    assert torch.equal(xf, want.view(blocks, N // blocks, cp, hb, wb).permute(0, 2, 3, 4, 1).contiguous())
    ws = ops.w_s2d_tap_major(w, s)
    assert ws.shape == (2, Cout, m * m, cp)
    wpad = torch.zeros(2, Cout, C, s * m, s * m, device="cuda")
    wpad[..., :k, :k] = w
    wwant = torch.zeros(2, Cout, m, m, cp, device="cuda")
    for c in range(C):
        for dy in range(s):
            for dx in range(s):
                wwant[..., (c * s + dy) * s + dx] = wpad[:, :, c, dy::s, dx::s]
    assert torch.equal(ws, wwant.view(2, Cout, m * m, cp))
    # the layer: every block's images against all draws' weights (input shared by the draws when blocks == 1)
    if blocks == 1:
        y = ops.c8s3_to_f32(ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1))   # [2, Cout, ho, wo, N]
        for e in range(2):
            xe, we, be = x.double().cpu().numpy(), w[e].double().cpu().numpy(), bias[e].double().cpu().numpy()
            ref = O.conv2d(xe, we, be, s, p, 1)
            mag = O.conv2d(np.abs(xe), np.abs(we), np.abs(be), s, p, 1)
            got = y[e].permute(3, 0, 1, 2).double().cpu().numpy()
            assert float((np.abs(got - ref) / mag).max()) <= TOL
        if ho % 2 == 0 and wo % 2 == 0:
            for act in (None, "softplus"):
                full = ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1, act=act)
                assert torch.equal(ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1, act=act, pool=True), ops.maxpool_c8s3(full, 2, 2))
        # block rows / columns that are all padding: declared to the launch, skipped, same bits (plain and pooled, every tile shape)
        zb = ops.s2d_zero_border(C, k, s, p, H, W)
        assert torch.equal(ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1, zero_border=zb), ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1))
        if ho % 2 == 0 and wo % 2 == 0:
            want = ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1, act="softplus", pool=True)
            for nt in (None, 3):
                assert torch.equal(ops.conv2d_c8x3_forward(xs, ws, bias, m, 1, 0, 1, act="softplus", pool=True, zero_border=zb, nt=nt), want)
    assert ops.s2d_zero_border(3, 11, 4, 5, 32, 32) == (1, 1, 0, 0)
    assert ops.s2d_layer_ok(3, 64, 11, 4, 5, 1, 32, 32) and not ops.s2d_layer_ok(3, 32, 5, 1, 2, 1, 32, 32)


def test_c8x3_is_exact_on_one_hot_weights(env):
    """hi + mid + lo == a exactly, through the c8 layout: a 1 x 1 convolution with a one-hot weight matrix returns its input bit
    for bit on channel scales 1e-20 .. 1e20."""
    ops = env["ops"]
    torch.manual_seed(3)
    C, B = 64, 256
    x = (torch.randn(1, C, 3, 3, B, device="cuda") * torch.logspace(-20, 20, C, device="cuda").view(1, C, 1, 1, 1)).contiguous()
    w = torch.eye(C, device="cuda").view(1, C, C, 1, 1).contiguous()
    y = ops.conv2d_c8x3_forward(ops.c8s3_from_f32(x), ops.w_tap_major(w), None, 1, 1, 0, 1)
    assert torch.equal(ops.c8s3_to_f32(y), x)


def test_c8x3_argument_errors(env):
    ops, L = env["ops"], __import__("bbb_hip")._lib
    x = ops.c8s3_from_f32(torch.randn(1, 32, 2, 2, 8, device="cuda"))
    with pytest.raises(L.BBBHipError):                       # 24 input channels: no 16-channel k step
        ops.conv2d_c8x3_forward(ops.c8s3_from_f32(torch.randn(1, 24, 2, 2, 8, device="cuda")), torch.randn(1, 8, 1, 24, device="cuda"), None, 1)
    with pytest.raises(L.BBBHipError):                       # the pooled form: no padding, even maps
        ops.conv2d_c8x3_forward(x, torch.randn(1, 8, 9, 32, device="cuda"), None, 3, 1, 1, 1, pool=True)
    with pytest.raises(L.BBBHipError):                       # 12 output channels in c8 form
        ops.conv2d_c8x3_forward(x, torch.randn(1, 12, 1, 32, device="cuda"), None, 1)
    y = ops.conv2d_c8x3_forward(x, torch.randn(1, 12, 1, 32, device="cuda"), None, 1, out_f32=True)
    assert y.shape == (1, 12, 2, 2, 8)


@pytest.mark.parametrize("draws", [1, 3, 17])
def test_parameter_pass_writes_tap_major_rows(env, draws):
    """bbb_segment_t::w_tm_cin: the sampled weights of flagged conv tensors come out tap-major, bit for bit the transposition of
    what the dense launch writes (the noise of a weight is keyed by its index in the tensor's OWN order), dense tensors and the KL
    scalar are untouched -- for 5 x 5, 3 x 3 and rectangular taps, row counts that leave a ragged last block, and enough elements
    that a block's (row, channel) range straddles rows."""
    ops = env["ops"]
    torch.manual_seed(draws)
    shapes = [(24, 64, 5, 5), (7,), (40, 192, 3, 3), (40,), (9, 8, 2, 3), (10, 128), (3, 16, 1, 7)]
    mus = [torch.randn(*s, device="cuda") * 0.1 for s in shapes]
    rhos = [torch.randn(*s, device="cuda") * 0.1 - 5.0 for s in shapes]
    ids = list(range(len(shapes)))
    tm = [len(s) == 4 for s in shapes]
    kl_d, ws_d = ops.sample_weights(mus, rhos, 0.0, 0.1, ids, 11, 5, draws)
    kl_t, ws_t = ops.sample_weights_tm(mus, rhos, 0.0, 0.1, ids, 11, 5, draws, tm)
    assert torch.equal(kl_d, kl_t)
    for wd, wt, flag in zip(ws_d, ws_t, tm):
        if flag:
            assert torch.equal(wt, ops.w_tap_major(wd))
        else:
            assert torch.equal(wt, wd)


import ref_port_torch as P


@pytest.mark.parametrize("model,shape,classes", [("alexnet", (512, 3, 32, 32), 10), ("3conv3fc", (256, 3, 32, 32), 10),
                                                 ("lenet", (512, 1, 32, 32), 10)])
def test_c8_chain_model_step(env, model, shape, classes):
    """The whole Monte-Carlo step in split-bf16 mode with the layers behind the first one on the MFMA-ready-operand kernel
    (LaunchConfig.c8x3, the default) against the same mode on round 4's kernel and against the fp32 path, same noise: KL bit for
    bit (the tap-major parameter pass leaves sigma / KL to the ordinary chunks), log-probabilities within 1e-5 of their largest
    magnitude; eager = hipGraph replay bit for bit.  3Conv3FC: its flatten cuts 2 x 2 maps into features (the chain passes through
    fp32 there); LeNet: only fc1 (400 features) qualifies -- one launch between two conversions."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel(model, shape[1], classes, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(*shape, device="cuda")
    E = 10
    with torch.no_grad():
        env["rng"].manual_seed(3, call=0)
        lo32, kl32 = ens.mc_forward(net, x, E)
        with ops.use_config(gemm_mode="bf16x3", c8x3=False):
            env["rng"].manual_seed(3, call=0)
            lo_old, kl_old = ens.mc_forward(net, x, E)
        with ops.use_config(gemm_mode="bf16x3"):
            assert ops.current_config().c8x3
            env["rng"].manual_seed(3, call=0)
            lo, kl = ens.mc_forward(net, x, E)
            env["rng"].manual_seed(3, call=0)
            g = ens.GraphedMC(net, x, E)
            lo_g, kl_g = g.step()
            torch.cuda.synchronize()
    assert torch.equal(kl, kl32) and torch.equal(kl, kl_old) and torch.equal(kl_g, kl)
    assert torch.equal(lo_g, lo)
    scale = float(lo32.abs().max())
    assert float((lo - lo32).abs().max()) <= 1e-5 * scale and float((lo - lo_old).abs().max()) <= 1e-5 * scale
    assert not torch.equal(lo, lo_old)                      # the new kernel really ran


def test_c8_chain_partitions_are_the_whole_step(env):
    """Work units of a sharded step and several steps per launch on the c8 chain: bit for bit the corresponding slices of the whole
    step (same MFMA sequence per output element whatever the launch holds; the first layer runs in space-to-depth form, its
    block image cut per batch slice / per step)."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    E, S = 10, 2
    with torch.no_grad(), ops.use_config(gemm_mode="bf16x3"):
        full, kl = ens._mc_logits_chwn(net, x, E, 7, 3)                          # [E, C, B]
        for rank in (0, 3):
            lo, hi = ens.unit_range(E, S, rank, 4)
            part, klp = ens._mc_logits_chwn(net, x, E, 7, 3, units=(S, lo, hi))  # [hi-lo, C, B/S]
            assert torch.equal(klp, kl)
            for i, u in enumerate(range(lo, hi)):
                j, sl = divmod(u, S)
                assert torch.equal(part[i], full[j][:, sl * 256:(sl + 1) * 256])
        x2 = torch.cat([x, torch.rand(512, 3, 32, 32, device="cuda")])
        two, _ = ens._mc_logits_chwn(net, x2, E, 7, 3, groups=2)                # [2 E, C, B]: step 1 under calls 3 + E ..
        assert torch.equal(two[:E], full)
        second, _ = ens._mc_logits_chwn(net, x2[512:], E, 7, 3 + E)
        assert torch.equal(two[E:], second)


# ---- the LRT form (bbb_lrt_conv2d_c8x3_fwd): both contractions on the 16-bit matrix pipe, the fp32 LRT kernel's noise ----
LRT_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, E, x_shared, act
    (256, 64, 4, 4, 192, 5, 1, 2, 2, False, "softplus"),     # AlexNet conv2
    (132, 32, 5, 7, 72, 3, 1, 1, 3, True, "relu"),           # ragged tiles, one input slab for three draws
    (64, 128, 1, 1, 16, 1, 1, 0, 2, False, None),            # linear
    (40, 48, 6, 6, 24, 3, 2, 0, 1, False, "softplus"),       # 48 channels, stride 2
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,E,xs,act", LRT_CASES)
@pytest.mark.parametrize("out_f32", [False, True])
def test_lrt_c8x3_launch_vs_the_fp32_lrt_kernel(env, B, Cin, H, W, Cout, k, s, p, E, xs, act, out_f32):
    """Same layer, same noise elements (stream (seed, call0 + draw, stream_id), canonical output index): the two kernels differ by
    the rounding of two contractions only -- outputs within 2e-5 of the largest magnitude; the six-plane output's squares are the
    fp32 squares of its values; sample=False returns act(act_mu)."""
    ops = env["ops"]
    torch.manual_seed(B + Cout)
    x = torch.randn(1 if xs else E, Cin, H, W, B, device="cuda")
    w_mu = torch.randn(Cout, Cin, k, k, device="cuda") * 0.1
    w_var = torch.rand(Cout, Cin, k, k, device="cuda") * 0.01
    b_mu = torch.randn(Cout, device="cuda") * 0.1
    b_var = torch.rand(Cout, device="cuda") * 0.01
    seed, call0, sid = 77, 5, 6
    ref, _, _ = ops.lrt_conv2d_chwn_forward(x.expand(E, *x.shape[1:]).contiguous(), w_mu, w_var, b_mu, b_var, seed, call0, sid, s, p, 1, act=act)
    x6 = ops.c8s3_from_f32(x, squares=True)
    wm, wv = ops.w_tap_major(w_mu.unsqueeze(0))[0], ops.w_tap_major(w_var.unsqueeze(0))[0]
    got6 = ops.lrt_conv2d_c8x3_forward(x6, wm, wv, b_mu, b_var, k, seed, call0, sid, s, p, 1, act=act, out_f32=out_f32, n_slabs=E)
    got = got6 if out_f32 else ops.c8s3_to_f32(got6)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    if not out_f32:
        assert got6.shape[1] == 6
        sq = ops.c8s3_to_f32(torch.cat([got6[:, 3:], got6[:, 3:]], dim=1))       # (the squares' planes read as values)
        assert torch.equal(sq, got * got)
        # pooling six-plane slabs: maximum of the values, squares of the pooled values
        if got6.shape[3] >= 2 and got6.shape[4] >= 2:
            pl = ops.maxpool_c8s3(got6, 2, 2)
            pv = ops.maxpool_chwn(got, 2, 2)
            assert torch.equal(ops.c8s3_to_f32(pl), pv)
            assert torch.equal(ops.c8s3_to_f32(torch.cat([pl[:, 3:], pl[:, 3:]], dim=1)), pv * pv)
    det, _, _ = ops.lrt_conv2d_chwn_forward(x.expand(E, *x.shape[1:]).contiguous(), w_mu, w_var, b_mu, b_var, seed, call0, sid, s, p, 1, act=act,
                                            sample=False)
    got_d = ops.lrt_conv2d_c8x3_forward(x6, wm, wv, b_mu, b_var, k, seed, call0, sid, s, p, 1, act=act, out_f32=True, n_slabs=E, sample=False)
    assert float((got_d - det).abs().max()) <= 2e-5 * float(det.abs().max())


def test_lrt_c8x3_noise_follows_the_global_image_index(env):
    """b_offset (batch-parallel shards) and work units key the noise by the GLOBAL image: a shard's / a unit's launch is bit for bit
    the corresponding slice of the whole launch."""
    ops = env["ops"]
    torch.manual_seed(2)
    E, Cin, H, W, B, Cout = 2, 32, 3, 3, 64, 40
    x = torch.randn(E, Cin, H, W, B, device="cuda")
    wm = ops.w_tap_major(torch.randn(1, Cout, Cin, 3, 3, device="cuda") * 0.1)[0]
    wv = ops.w_tap_major(torch.rand(1, Cout, Cin, 3, 3, device="cuda") * 0.01)[0]
    b_mu, b_var = torch.randn(Cout, device="cuda") * 0.1, torch.rand(Cout, device="cuda") * 0.01
    full = ops.c8s3_to_f32(ops.lrt_conv2d_c8x3_forward(ops.c8s3_from_f32(x, squares=True), wm, wv, b_mu, b_var, 3, 9, 4, 2, 1, 1, 1, act="softplus"))
    half = ops.c8s3_to_f32(ops.lrt_conv2d_c8x3_forward(ops.c8s3_from_f32(x[..., 32:].contiguous(), squares=True), wm, wv, b_mu, b_var, 3, 9, 4, 2,
                                                       1, 1, 1, act="softplus", b_offset=32))
    assert torch.equal(half, full[..., 32:])
    # units: S = 2 slices per draw, units 1..3 of the (draw, slice) grid = (0, 1), (1, 0), (1, 1)
    xu = torch.stack([x[0][..., 32:], x[1][..., :32], x[1][..., 32:]]).contiguous()
    un = ops.c8s3_to_f32(ops.lrt_conv2d_c8x3_forward(ops.c8s3_from_f32(xu, squares=True), wm, wv, b_mu, b_var, 3, 9, 4, 2, 1, 1, 1, act="softplus",
                                                     units=(2, 1), n_units=3))
    assert torch.equal(un[0], full[0][..., 32:]) and torch.equal(un[1], full[1][..., :32]) and torch.equal(un[2], full[1][..., 32:])


@pytest.mark.parametrize("model,shape,classes,E", [("alexnet", (512, 3, 32, 32), 100, 1), ("alexnet", (256, 3, 32, 32), 10, 4),
                                                   ("3conv3fc", (64, 3, 32, 32), 10, 2)])
def test_c8_chain_lrt_model_step(env, model, shape, classes, E):
    """LRT models in split-bf16 mode: every layer behind the first (and AlexNet's conv1 in space-to-depth form) on the LRT form of
    the MFMA-ready-operand kernel, six-plane slabs between them -- against the fp32 LRT path under the same noise: KL bit for bit,
    log-probabilities within 2e-5 of their largest magnitude; eager = hipGraph replay; a rank's work units and several steps per
    launch are bit for bit the slices of the whole step."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(1)
    net = env["zoo"].getModel(model, shape[1], classes, P.CONFIG_PRIORS, "lrt", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(*shape, device="cuda")
    with torch.no_grad():
        env["rng"].manual_seed(3, call=0)
        lo32, kl32 = ens.mc_forward(net, x, E)
        with ops.use_config(gemm_mode="bf16x3", c8x3=False):
            env["rng"].manual_seed(3, call=0)
            lo_old, _ = ens.mc_forward(net, x, E)
        assert torch.equal(lo_old, lo32)                     # (without the chain LRT layers keep their fp32 kernel in the mode)
        with ops.use_config(gemm_mode="bf16x3"):
            env["rng"].manual_seed(3, call=0)
            lo, kl = ens.mc_forward(net, x, E)
            env["rng"].manual_seed(3, call=0)
            g = ens.GraphedMC(net, x, E)
            lo_g, kl_g = g.step()
            torch.cuda.synchronize()
            assert torch.equal(kl, kl32) and torch.equal(kl_g, kl) and torch.equal(lo_g, lo)
            scale = float(lo32.abs().max())
            assert float((lo - lo32).abs().max()) <= 2e-5 * scale and not torch.equal(lo, lo32)
            # partitions: work units of a sharded step, several steps per launch
            full, _ = ens._mc_logits_chwn(net, x, E, 7, 3)
            S = 2
            lo_u, hi_u = ens.unit_range(E, S, 1, 2)
            part, _ = ens._mc_logits_chwn(net, x, E, 7, 3, units=(S, lo_u, hi_u))
            for i, u in enumerate(range(lo_u, hi_u)):
                j, sl = divmod(u, S)
                bs = shape[0] // S
                assert torch.equal(part[i], full[j][:, sl * bs:(sl + 1) * bs])
            x2 = torch.cat([x, torch.rand(*shape, device="cuda")])
            two, _ = ens._mc_logits_chwn(net, x2, E, 7, 3, groups=2)
            assert torch.equal(two[:E], full)
            second, _ = ens._mc_logits_chwn(net, x2[shape[0]:], E, 7, 3 + E)
            assert torch.equal(two[E:], second)


def test_c8x3_random_geometries(env):
    """60 seeded random layer geometries (channels 16 .. 144, maps 1 .. 11, kernels 1 .. 5, stride / dilation 1 .. 3, padding 0 .. 3,
    batches that leave ragged image tiles, shared and per-draw inputs): the BBB form against the fp32 kernel (2e-5 of the largest
    magnitude), its pooled form against conv + pool (bit for bit) wherever it applies, the LRT form against the fp32 LRT kernel under
    the same noise."""
    ops = env["ops"]
    rs = np.random.RandomState(2024)
    done = pooled = 0
    while done < 60:
        Cin = int(rs.choice([16, 32, 48, 64, 80, 144]))
        Cout = int(rs.choice([8, 24, 64, 72, 136]))
        H, W = int(rs.randint(1, 12)), int(rs.randint(1, 12))
        kh = int(rs.randint(1, 6))
        k = (kh, kh)
        s, d, p = int(rs.randint(1, 4)), int(rs.randint(1, 4)), int(rs.randint(0, 4))
        ho = (H + 2 * p - d * (kh - 1) - 1) // s + 1
        wo = (W + 2 * p - d * (kh - 1) - 1) // s + 1
        if ho <= 0 or wo <= 0 or d * (kh - 1) < p:
            continue
        B = int(rs.choice([4, 36, 132, 260]))
        E = int(rs.randint(1, 4))
        xs = bool(rs.randint(0, 2))
        torch.manual_seed(done)
        x = torch.randn(1 if xs else E, Cin, H, W, B, device="cuda")
        w = torch.randn(E, Cout, Cin, kh, kh, device="cuda") * (1.0 / (Cin * kh * kh) ** 0.5)
        b = torch.randn(E, Cout, device="cuda") * 0.1
        ref = ops.conv2d_chwn_forward(x, w, b, s, p, d, act="softplus", bf16x3=False)
        xc, wt = ops.c8s3_from_f32(x), ops.w_tap_major(w)
        got6 = ops.conv2d_c8x3_forward(xc, wt, b, k, s, p, d, act="softplus")
        got = ops.c8s3_to_f32(got6)
        assert got.shape == ref.shape and float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (Cin, Cout, H, W, kh, s, d, p, B, E, xs)
        if p == 0 and ho % 2 == 0 and wo % 2 == 0:
            assert torch.equal(ops.conv2d_c8x3_forward(xc, wt, b, k, s, p, d, act="softplus", pool=True), ops.maxpool_c8s3(got6, 2, 2))
            pooled += 1
        # LRT form: one (mu, sigma^2) pair for all slabs
        w_var = torch.rand(Cout, Cin, kh, kh, device="cuda") * (0.01 / (Cin * kh * kh))
        b_var = torch.rand(Cout, device="cuda") * 1e-3
        xe = x.expand(E, *x.shape[1:]).contiguous()
        lref, _, _ = ops.lrt_conv2d_chwn_forward(xe, w[0], w_var, b[0], b_var, 5, 9, 6, s, p, d, act="relu")
        lgot = ops.c8s3_to_f32(ops.lrt_conv2d_c8x3_forward(ops.c8s3_from_f32(x, squares=True), ops.w_tap_major(w[:1])[0],
                                                           ops.w_tap_major(w_var.unsqueeze(0))[0], b[0], b_var, k, 5, 9, 6, s, p, d, act="relu",
                                                           n_slabs=E))
        assert float((lgot - lref).abs().max()) <= 2e-5 * max(float(lref.abs().max()), 1e-3), ("lrt", Cin, Cout, H, W, kh, s, d, p, B, E, xs)
        done += 1
    assert pooled >= 3
