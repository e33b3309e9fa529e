"""Split-fp16 contraction (bbb_conv2d_chwn_f16x2_fwd, ops.gemm_mode = "fp16x2"): fp32 operands cut into two fp16 pieces while
staged, three products per fp32 product on the 16-bit matrix pipe, fp32 accumulation.  Held to the SAME bound as the fp32 kernel
against the float64 oracle (tests/test_gpu_splitk.py: 4e-6 of sum_k |w||x|), and measured against it.  Run with -m gpu."""
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
def f16x2(env):
    env["ops"].gemm_mode = "fp16x2"
    yield
    env["ops"].gemm_mode = "fp32"


@pytest.fixture()
def every_launch(env):
    """No launch-size policy: also the small test shapes run on the split-fp16 kernel (and no fp32 split contraction)."""
    ops = env["ops"]
    keep = (ops.f16x2_min_workgroups, ops.split_k)
    ops.f16x2_min_workgroups, ops.split_k = 0, False
    ops._split_plans.clear()
    yield
    ops.f16x2_min_workgroups, ops.split_k = keep
    ops._split_plans.clear()


CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, E, x_shared
    (512, 3, 32, 32, 64, 11, 4, 5, 1, 2, True),      # AlexNet conv1
    (256, 64, 4, 4, 192, 5, 1, 2, 1, 2, False),      # AlexNet conv2 shape: most taps of border pixels out of bounds
    (132, 6, 9, 7, 70, 3, 1, 1, 1, 2, False),        # ragged image tile, ragged channel tile
    (8, 16, 6, 6, 130, 3, 2, 1, 2, 3, False),        # stride + dilation
    (40, 520, 1, 1, 10, 1, 1, 0, 1, 2, False),       # linear, K = 520 (three table chunks)
    (64, 256, 2, 2, 256, 3, 1, 1, 1, 1, True),       # AlexNet conv4 shape, one draw: the 128-image tile
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p,d,E,xs", CASES)
@pytest.mark.parametrize("xscale,wscale,bound", [(3.0, 0.2, TOL), (0.25, 0.01, TOL), (100.0, 8.0, TOL), (0.004, 0.0003, 2e-4)])
def test_f16x2_launch_vs_oracle_and_fp32_kernel(env, f16x2, every_launch, B, Cin, H, W, Cout, k, s, p, d, E, xs, xscale, wscale, bound):
    """Inside the operand window (pconv_f16x2.cuh) the fp32 kernel's bound holds -- at its lower edge (Gaussian tensors of
    typical size 0.25 / 0.01: a few per cent of their elements lose the lo piece) with 1e-6 instead of 2e-7; the last scale pair
    lies BELOW the window (most lo pieces are fp16 subnormals, which the matrix instruction flushes): graceful, stated
    degradation, no garbage."""
    ops = env["ops"]
    torch.manual_seed(B + Cout)
    x = torch.randn(1 if xs else E, Cin, H, W, B, device="cuda") * xscale
    w = torch.randn(E, Cout, Cin, k, k, device="cuda") * wscale
    bias = torch.randn(E, Cout, device="cuda") * (xscale * wscale)
    y = ops.conv2d_chwn_forward(x, w, bias, s, p, d, act=None, f16x2=True)
    ops.gemm_mode = "fp32"
    y32 = ops.conv2d_chwn_forward(x, w, bias, s, p, d, act=None)
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
    print(f"relative to sum|w||x|: split-fp16 {worst:.2e}, fp32 kernel {worst32:.2e}")
    assert worst <= bound, (worst, worst32)
    if xscale >= 3.0:
        assert worst <= 1.5 * worst32 + 1e-7          # operands of O(0.1-1) and larger: not a worse approximation than the fp32 kernel


def test_f16x2_model_step_matches_fp32_step(env, f16x2):
    """The whole 512 x 10 AlexNet step in both modes, same noise: log-probabilities agree to 1e-5 of their largest magnitude (the
    fp32 path itself sits 2.4e-6 of max|logit| from the float64 oracle at this size), KL identical."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    with torch.no_grad():
        env["rng"].manual_seed(3, call=0)
        lo, kl = ens.mc_forward(net, x, 10)
        ops.gemm_mode = "fp32"
        env["rng"].manual_seed(3, call=0)
        lo32, kl32 = ens.mc_forward(net, x, 10)
    assert torch.equal(kl, kl32)
    assert float((lo - lo32).abs().max()) <= 1e-5 * float(lo32.abs().max())


def test_f16x2_work_units_match_the_whole_step(env, f16x2):
    """A rank's (draw x batch-slice) work units in split-fp16 mode: each unit's logits equal the corresponding block of the
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


def test_f16x2_activation_scale_follows_the_data(env, f16x2, every_launch):
    """Activations far below the fixed-scale window (typical 0.004): with the launch told max|x| (amax_in) the fp32 bound holds
    again, and the launch publishes max|y| (amax_out) for the next layer."""
    ops = env["ops"]
    torch.manual_seed(5)
    B, Cin, H, W, Cout, k = 256, 64, 4, 4, 192, 5
    x = torch.randn(1, Cin, H, W, B, device="cuda") * 0.004
    w = torch.randn(2, Cout, Cin, k, k, device="cuda") * 0.2
    bias = torch.randn(2, Cout, device="cuda") * 1e-3
    amax_in = torch.zeros(ops.AMAX_SLOTS, device="cuda")
    amax_in[5] = x.abs().max()                                    # the bound is the maximum over the slots
    amax_out = torch.zeros(ops.AMAX_SLOTS, device="cuda")
    y_fix = ops.conv2d_chwn_forward(x, w, bias, 1, 2, 1, act=None, f16x2=True)
    y_dyn = ops.conv2d_chwn_forward(x, w, bias, 1, 2, 1, act=None, amax_in=amax_in, amax_out=amax_out, f16x2=True)
    assert float(amax_out.max()) == float(y_dyn.abs().max())
    errs = {}
    for name, y in (("fixed", y_fix), ("dynamic", y_dyn)):
        worst = 0.0
        for e in range(2):
            xe = x[0].permute(3, 0, 1, 2).double().cpu().numpy()
            we, be = w[e].double().cpu().numpy(), bias[e].double().cpu().numpy()
            want = O.conv2d(xe, we, be, 1, 2, 1)
            mag = O.conv2d(np.abs(xe), np.abs(we), np.abs(be), 1, 2, 1)
            worst = max(worst, float((np.abs(y[e].permute(3, 0, 1, 2).double().cpu().numpy() - want) / mag).max()))
        errs[name] = worst
    print(errs)
    assert errs["dynamic"] <= TOL and errs["fixed"] > errs["dynamic"]


def test_f16x2_model_with_tiny_inputs(env, f16x2):
    """The ensemble path feeds every layer the previous layer's max|y|: a 512 x 10 step on images scaled by 1e-3 (far below the
    fixed activation window) still agrees with the fp32 path to 1e-5."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda") * 1e-3
    with torch.no_grad():
        env["rng"].manual_seed(3, call=0)
        lo, kl = ens.mc_forward(net, x, 10)
        ops.gemm_mode = "fp32"
        env["rng"].manual_seed(3, call=0)
        lo32, kl32 = ens.mc_forward(net, x, 10)
    assert float((lo - lo32).abs().max()) <= 1e-5 * float(lo32.abs().max())


def test_f16x2_mode_leaves_training_on_the_fp32_kernel(env, f16x2):
    """Gradients are 1e-6-sized operands, far below the split's operand window: the training path (forward and role-swapped
    backward launches of fast_train) ignores the mode -- same gradients, bit for bit, as in fp32 mode."""
    import torch.nn.functional as F
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (512,), device="cuda")
    grads = {}
    for mode in ("fp16x2", "fp32"):
        ops.gemm_mode = mode
        net.zero_grad(set_to_none=True)
        env["rng"].manual_seed(5, call=0)
        lo, kl = ens.mc_forward(net, x, 4, kl_mode="mean")
        assert ens.stats["path"] == "chwn-autograd"
        (F.nll_loss(lo, y) * 50000.0 + 0.1 * kl).backward()
        grads[mode] = [p.grad.detach().clone() for p in net.parameters()]
        del lo, kl
    for a, b in zip(grads["fp16x2"], grads["fp32"]):
        assert torch.equal(a, b)


def test_precision_argument_selects_the_mode(env):
    """precision="fp16x2" on the ensemble entry points == ops.gemm_mode = "fp16x2" (same bits), graph replay included."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    with torch.no_grad():
        env["rng"].manual_seed(3, call=0)
        a, _ = ens.mc_forward(net, x, 10, precision="fp16x2")
        env["rng"].manual_seed(3, call=0)
        g = ens.GraphedMC(net, x, 10, precision="fp16x2")
        b, _ = g.step()
        torch.cuda.synchronize()
        ops.gemm_mode = "fp16x2"
        try:
            env["rng"].manual_seed(3, call=0)
            c, _ = ens.mc_forward(net, x, 10)
        finally:
            ops.gemm_mode = "fp32"
        env["rng"].manual_seed(3, call=0)
        d, _ = ens.mc_forward(net, x, 10)
    assert torch.equal(a, b) and torch.equal(a, c) and not torch.equal(a, d)


def test_f16x2_weight_scale_follows_the_parameters(env, f16x2):
    """Weights far below the fixed-scale window (posterior means of 1e-4): the layer's analytic bound max(|mu| + 6.66 sigma) sets
    the weight scale, the step agrees with the fp32 path to 1e-5; after an in-place parameter update a captured step picks the new
    bound up at its next replay."""
    ens, ops = env["ens"], env["ops"]
    torch.manual_seed(0)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    with torch.no_grad():
        for l in ens.bayesian_layers(net):
            l.W_mu.mul_(1e-3)
            l.W_rho.fill_(-12.0)
    x = torch.rand(512, 3, 32, 32, device="cuda")
    with torch.no_grad():
        env["rng"].manual_seed(3, call=0)
        g = ens.GraphedMC(net, x, 10)
        lo, _ = g.step()
        torch.cuda.synchronize()
        lo = lo.clone()
        b0 = float(ens._weight_bound(ens.bayesian_layers(net)[0])[0])
        ops.gemm_mode = "fp32"
        env["rng"].manual_seed(3, call=0)
        lo32, _ = ens.mc_forward(net, x, 10)
        ops.gemm_mode = "fp16x2"
        assert float((lo - lo32).abs().max()) <= 1e-5 * float(lo32.abs().max())
        for l in ens.bayesian_layers(net):
            l.W_mu.mul_(100.0)                                   # an optimizer step, exaggerated
        env["rng"].manual_seed(3, call=10)
        lo2, _ = g.step()
        torch.cuda.synchronize()
        b1 = float(ens._weight_bound(ens.bayesian_layers(net)[0])[0])
        assert b1 > 50 * b0
        ops.gemm_mode = "fp32"
        env["rng"].manual_seed(3, call=10)
        lo2_32, _ = ens.mc_forward(net, x, 10)
        assert float((lo2 - lo2_32).abs().max()) <= 1e-5 * float(lo2_32.abs().max())
