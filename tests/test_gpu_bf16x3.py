"""Range-free split-bf16 contraction (bbb_conv2d_chwn_bf16x3_fwd, ops.gemm_mode = "bf16x3"): fp32 operands cut into three bf16
pieces while staged (exact: 3 x 8 significand bits, fp32's exponent range), six products per fp32 product on the 16-bit matrix
pipe, fp32 accumulation.  Held to the SAME bound as the fp32 kernel against the float64 oracle (4e-6 of sum_k |w||x|) on every
operand scale -- there is no operand window -- and measured against it.  Run with -m gpu."""
import numpy as np
import pytest
import torch

import bbb_numpy as O
import ref_port_torch as P

pytestmark = pytest.mark.gpu
TOL = 4e-6


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo)


@pytest.fixture()
def bf16x3(env):
    env["ops"].gemm_mode = "bf16x3"
    yield
    env["ops"].gemm_mode = "fp32"


@pytest.fixture()
def every_launch(env):
    """No launch-size policy: also the small test shapes run on the split-bf16 kernel."""
    ops = env["ops"]
    keep = ops.bf16x3_min_workgroups
    ops.bf16x3_min_workgroups = 0
    yield
    ops.bf16x3_min_workgroups = keep


CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, E, x_shared
    (512, 3, 32, 32, 64, 11, 4, 5, 1, 2, True),      # AlexNet conv1
    (256, 64, 4, 4, 192, 5, 1, 2, 1, 2, False),      # AlexNet conv2 shape: most taps of border pixels out of bounds
    (132, 6, 9, 7, 70, 3, 1, 1, 1, 2, False),        # ragged image tile, ragged channel tile
    (8, 16, 6, 6, 130, 3, 2, 1, 2, 3, False),        # stride + dilation
    (40, 520, 1, 1, 10, 1, 1, 0, 1, 2, False),       # linear, K = 520 (three table chunks)
    (64, 256, 2, 2, 256, 3, 1, 1, 1, 1, True),       # AlexNet conv4 shape, one draw
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,d,E,xs", CASES)
@pytest.mark.parametrize("xscale,wscale", [(3.0, 0.2), (0.004, 0.0003), (1e-6, 1e-9), (1e6, 1e-12), (1e-15, 1e15)])
def test_bf16x3_launch_vs_oracle_and_fp32_kernel(env, bf16x3, every_launch, B, Cin, H, W, Cout, k, s, p, d, E, xs, xscale, wscale):
    """The fp32 kernel's bound on every operand scale, O(1) activations and weights as well as 1e-9-sized variances, 1e-6-sized
    gradients and 1e6-sized inputs (round 3's split-fp16 form was 5e-3 / 0.3 off on the last three: its operand window)."""
    ops = env["ops"]
    torch.manual_seed(B + Cout)
    x = torch.randn(1 if xs else E, Cin, H, W, B, device="cuda") * xscale
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * wscale
    bias = torch.randn(E, Cout, device="cuda") * (xscale * wscale)
    y = ops.conv2d_chwn_forward(x, w, bias, s, p, d, act=None)
    y32 = ops.conv2d_chwn_forward(x, w, bias, s, p, d, act=None, bf16x3=False)
    assert not torch.equal(y, y32)                                  # the split kernel really ran
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
    print(f"relative to sum|w||x|: split-bf16 {worst:.2e}, fp32 kernel {worst32:.2e}")
    assert worst <= TOL, (worst, worst32)
    assert worst <= max(4e-7, 3.0 * worst32)                          # the fp32 kernel's class (measured 2.5-3.2e-7 vs 0.8-3.7e-7)


def test_bf16x3_split_is_exact(env):
    """hi + mid + lo == a exactly: a contraction against the identity returns the operand bit for bit (a 1 x 1 convolution
    with a one-hot weight matrix copies channels; any lost bit of a piece would show).  Channel scales 1e-20 .. 1e20; only below
    2^-110 ~ 1e-33, where the lo piece becomes a bf16 subnormal that the matrix instruction flushes, bits are lost."""
    ops = env["ops"]
    torch.manual_seed(3)
    C, B = 64, 256
    x = (torch.randn(1, C, 3, 3, B, device="cuda") * torch.logspace(-20, 20, C, device="cuda").view(1, C, 1, 1, 1)).contiguous()
    w = torch.eye(C, device="cuda").view(1, C, C, 1, 1).contiguous()
    y = ops.conv2d_chwn_forward(x, w, None, 1, 0, 1, act=None, bf16x3=True)
    saved, ops.bf16x3_min_workgroups = ops.bf16x3_min_workgroups, 0
    try:
        y = ops.conv2d_chwn_forward(x, w, None, 1, 0, 1, act=None, bf16x3=True)
    finally:
        ops.bf16x3_min_workgroups = saved
    assert torch.equal(y, x)


def test_bf16x3_model_step_matches_fp32_step(env, bf16x3):
    """The whole 512 x 10 AlexNet step in both modes, same noise: log-probabilities agree to 1e-5 of their largest magnitude (the
    fp32 path itself sits 2.4e-6 of max|logit| from the float64 oracle at this size), KL identical; also on images scaled by 1e-3
    (no activation scale to follow any more) and through the precision= argument, graph replay included."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    for scale in (1.0, 1e-3):
        x = torch.rand(512, 3, 32, 32, device="cuda") * scale
        with torch.no_grad():
            ops.gemm_mode = "bf16x3"
            env["rng"].manual_seed(3, call=0)
            lo, kl = ens.mc_forward(net, x, 10)
            ops.gemm_mode = "fp32"
            env["rng"].manual_seed(3, call=0)
            lo32, kl32 = ens.mc_forward(net, x, 10)
            env["rng"].manual_seed(3, call=0)
            lo_p, kl_p = ens.mc_forward(net, x, 10, precision="bf16x3")
            env["rng"].manual_seed(3, call=0)
            g = ens.GraphedMC(net, x, 10, precision="bf16x3")
            lo_g, kl_g = g.step()
            torch.cuda.synchronize()
        assert torch.equal(kl, kl32) and torch.equal(lo_p, lo) and torch.equal(lo_g, lo)
        assert not torch.equal(lo, lo32)
        assert float((lo - lo32).abs().max()) <= 1e-5 * float(lo32.abs().max())


def test_bf16x3_work_units_match_the_whole_step(env, bf16x3):
    """A rank's (draw x batch-slice) work units in split-bf16 mode: each unit's logits equal the corresponding block of the
    unsharded step (the shares' smaller launches may take the fp32 kernel -- launch-size policy in ops -- hence a tolerance)."""
    ens = env["ens"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    E, S = 10, 4
    with torch.no_grad():
        full, kl = ens._mc_logits_chwn(net, x, E, 7, 3)                          # [E, C, B]
        for rank in (0, 3, 7):
            lo, hi = ens.unit_range(E, S, rank, 8)
            part, klp = ens._mc_logits_chwn(net, x, E, 7, 3, units=(S, lo, hi))  # [hi-lo, C, B/S]
            assert torch.equal(klp, kl)
            for i, u in enumerate(range(lo, hi)):
                j, sl = divmod(u, S)
                want = full[j, :, sl * 128:(sl + 1) * 128]
                assert float((part[i] - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("lt", ["bbb", "lrt"])
def test_bf16x3_mode_covers_the_training_gemms(env, bf16x3, lt):
    """Range-free, so the mode also applies to the role-swapped gradient launches of the training path (1e-6-sized operands, which
    round 3's split-fp16 form could not take): gradients agree with the fp32 mode to 2e-5 of each tensor's largest entry -- the
    bound the fp32 path's own gradient checks use is 2e-3 -- and are not bit-identical (the split kernel really ran).  LRT layers:
    the forward keeps the fused fp32 kernel, the backward GEMMs take the mode."""
    import torch.nn.functional as F
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (512,), device="cuda")
    grads = {}
    for mode in ("bf16x3", "fp32"):
        ops.gemm_mode = mode
        net.zero_grad(set_to_none=True)
        env["rng"].manual_seed(5, call=0)
        lo, kl = ens.mc_forward(net, x, 4, kl_mode="mean")
        assert ens.stats["path"] == "chwn-autograd"
        (F.nll_loss(lo, y) * 50000.0 + 0.1 * kl).backward()
        grads[mode] = [p.grad.detach().clone() for p in net.parameters()]
        del lo, kl
    differ = 0
    for (n, _), a, b in zip(net.named_parameters(), grads["bf16x3"], grads["fp32"]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), n
        differ += int(not torch.equal(a, b))
    assert differ > 0


def test_s3_format_is_exact_and_transparent(env):
    """The split activation format S3 (three bf16 planes = the hi / mid / lo pieces of the same fp32 values): conversion is exact
    both ways; a launch that reads S3 and writes S3 computes, piece for piece, what the launch on the fp32 tensors computes (the
    in-kernel split IS the same function), so its result is bitwise the S3 form of that launch's fp32 output; pooling S3 planes
    is bitwise pooling the fp32 tensor."""
    ops = env["ops"]
    torch.manual_seed(4)
    E, Cin, H, W, B, Cout, k = 3, 64, 4, 4, 256, 192, 5
    x = torch.randn(E, Cin, H, W, B, device="cuda") * torch.logspace(-6, 6, Cin, device="cuda").view(1, Cin, 1, 1, 1)
    xs = ops.s3_from_f32(x)
    assert xs.shape == (E, 3, Cin, H, W, B) and xs.dtype == torch.bfloat16
    assert torch.equal(ops.s3_to_f32(xs), x)
    assert torch.equal(xs[:, 0].float() + xs[:, 1].float() + xs[:, 2].float(), x)
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * 0.05
    b = torch.randn(E, Cout, device="cuda")
    y = ops.conv2d_chwn_forward(x, w, b, 1, 2, 1, act="softplus", bf16x3=True)
    for xin, x_s3 in ((x, False), (xs, True)):
        for out_s3 in (False, True):
            got = ops.conv2d_chwn_forward(xin, w, b, 1, 2, 1, act="softplus", bf16x3=True, x_s3=x_s3, out_s3=out_s3)
            assert torch.equal(ops.s3_to_f32(got) if out_s3 else got, y), (x_s3, out_s3)
    # a shared input slab (first layer of an ensemble) and the x_div form
    y1 = ops.conv2d_chwn_forward(x[:1], w, b, 1, 2, 1, act=None, bf16x3=True)
    assert torch.equal(ops.conv2d_chwn_forward(xs[:1], w, b, 1, 2, 1, act=None, x_s3=True), y1)
    # pooling
    ys = ops.s3_from_f32(y)
    assert torch.equal(ops.s3_to_f32(ops.maxpool_chwn_s3(ys, 2, 2)), ops.maxpool_chwn(y, 2, 2))
    assert torch.equal(ops.s3_to_f32(ops.maxpool_chwn_s3(ys, 3, 1)), ops.maxpool_chwn(y, 3, 1))


@pytest.mark.parametrize("net_type,B,E", [("alexnet", 512, 10), ("3conv3fc", 256, 8), ("lenet", 64, 32)])
def test_s3_chain_equals_per_launch_splitting(env, bf16x3, every_launch, net_type, B, E):
    """A whole step in split-bf16 mode with the activations travelling as S3 between the layers (the default for steps of >=
    ops.s3_min_images rows) against the same step with every launch splitting its fp32 operands itself: same bits."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    cin = 1 if net_type == "lenet" else 3
    net = env["zoo"].getModel(net_type, cin, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, 32, 32, device="cuda")
    with torch.no_grad(), ops.use_config(c8x3=False):       # (round 4's kernel: the c8 chain of round 6 has its own tests, test_gpu_c8x3.py)
        chain, kl = ens._mc_logits_chwn(net, x, E, 7, 3)
        keep, ops.s3_min_images = ops.s3_min_images, 1 << 40
        try:
            plain, kl2 = ens._mc_logits_chwn(net, x, E, 7, 3)
        finally:
            ops.s3_min_images = keep
    assert torch.equal(kl, kl2) and torch.equal(chain, plain)
