"""Strong-scaling sharding of one MC step on ONE MI355X with simulated ranks (run with -m gpu): the (draw x batch-slice) work
units of SURVEY.md 8(e).  Every "rank" runs exactly the launches a real rank would (same unit range, same weight sets, same
kernels with the unit -> (weight set, input slab) mapping) on the shared device; the combine step is the arithmetic of
ensemble.combine_ranks.  The real collective is covered by the gloo tests in test_host_cpu.py.
Reference behaviour to match: the N-rank result equals the 1-rank result (main_bayesian.py:73-80 has no multi-GPU path)."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

import ref_port_torch as P

pytestmark = pytest.mark.gpu          # bitwise comparisons across launch sizes, at the SHIPPED defaults (ops.split_k on)


@pytest.fixture(scope="module")
def env():
    import layers
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(layers=layers, ops=ops, rng=rng, ens=ensemble, zoo=zoo)


def build(env, net_type, lt, ncls, seed=0):
    torch.manual_seed(seed)
    net = env["zoo"].getModel(net_type, 3, ncls, P.CONFIG_PRIORS, lt, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    return net


@pytest.mark.parametrize("net_type,lt,ncls,B,E,world,precision", [
    ("alexnet", "bbb", 10, 512, 10, 8, "fp32"),      # the metric config on 8 ranks: 40 quarter-batch units, 5 per rank
    ("alexnet", "bbb", 10, 512, 10, 4, "fp32"),      # half-batch units
    ("alexnet", "bbb", 10, 512, 10, 3, "fp32"),      # uneven deal: 7 / 7 / 6 half-batch units
    ("alexnet", "bbb", 10, 512, 25, 8, "fp32"),      # configs[3]: num_ens = 25 over 8 ranks
    ("alexnet", "lrt", 100, 512, 10, 8, "fp32"),     # LRT: activation noise keyed by (draw, GLOBAL image index)
    ("3conv3fc", "bbb", 10, 256, 10, 8, "bf16"),     # bf16 storage path, 64-image slices
    ("lenet", "bbb", 10, 256, 3, 8, "fp32"),         # more ranks than units: two ranks idle
])
def test_unit_sharding_matches_single_device(env, net_type, lt, ncls, B, E, world, precision):
    ens, ops = env["ens"], env["ops"]
    cin = 1 if net_type == "lenet" else 3
    torch.manual_seed(0)
    net = env["zoo"].getModel(net_type, cin, ncls, P.CONFIG_PRIORS, lt, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    x = torch.rand(B, cin, 32, 32, device="cuda")
    seed, call0 = 77, 9
    mult = 8 if precision == "bf16" else 4
    S = ens.plan_slices(E, world, B, multiple=mult)
    assert S > 1
    Bs = B // S
    with torch.no_grad():
        full, kl_full = ens._mc_logits_chwn(net, x, E, seed, call0, precision=precision)       # [E, C, B]
        want = ops.mc_tail_cb(full, mean_over=E)
        blocks, kl_sum, covered = [], 0.0, 0
        for r in range(world):
            lo, hi = ens.unit_range(E, S, r, world)
            if hi == lo:
                blocks.append(torch.full((B, ncls), -float("inf"), device="cuda"))
                continue
            logits, kl1 = ens._mc_logits_chwn(net, x, E, seed, call0, precision=precision, units=(S, lo, hi))
            assert tuple(logits.shape) == (hi - lo, ncls, Bs)
            for e, u in enumerate(range(lo, hi)):                       # every unit = the matching slice of the matching draw
                j, s = divmod(u, S)
                ref = full[j][:, s * Bs:(s + 1) * Bs]
                if precision == "bf16":
                    # the bf16 launcher picks tile shape and k-groups by launch size (a rank's share is a smaller launch): the
                    # fp32 summation order differs, a hidden activation on a rounding boundary flips by one bf16 ulp
                    assert float((logits[e] - ref).abs().max()) <= 2e-2 * float(ref.abs().max()), (r, u)
                else:
                    assert torch.equal(logits[e], ref), (r, u)
            covered += hi - lo
            assert kl1.item() == kl_full.item()
            lse, _ = ens._local_lse(net, x, E, seed, call0, 0, precision=precision, units=(S, lo, hi))
            blocks.append(lse)
            kl_sum += kl1.item() * (hi - lo) / S
        assert covered == E * S
        got = torch.logsumexp(torch.stack(blocks), 0) - math.log(E)
    # log-probabilities reach -150 here: 1 fp32 ulp = 1.5e-5; the only difference is the order of the log-sum-exp
    if precision == "bf16":
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-2 * float(want.abs().max()))
    else:
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=5e-7, atol=3e-6)
    assert abs(kl_sum - E * kl_full.item()) <= 1e-6 * E * kl_full.item()


def test_mc_tail_units_kernel(env):
    """bbb_mc_tail_units against torch: per-slice log-sum-exp over the local units, -inf where a slice has none."""
    ops = env["ops"]
    g = torch.Generator(device="cuda").manual_seed(3)
    S, Bs, C = 4, 24, 7
    for off, U in [(0, 8), (1, 5), (3, 2), (2, 1)]:
        logits = torch.randn(U, C, Bs, device="cuda", generator=g) * 4
        got = ops.mc_tail_units(logits, S, off, mean_over=0)
        want = torch.full((S * Bs, C), -float("inf"), device="cuda")
        for e in range(U):
            s = (off + e) % S
            ls = torch.log_softmax(logits[e].t(), dim=1)            # [Bs, C]
            want[s * Bs:(s + 1) * Bs] = torch.logaddexp(want[s * Bs:(s + 1) * Bs], ls)
        assert torch.equal(torch.isinf(got), torch.isinf(want))
        fin = ~torch.isinf(want)
        np.testing.assert_allclose(got[fin].cpu().numpy(), want[fin].cpu().numpy(), rtol=0, atol=3e-6)


def test_batch_parallel_shards_equal_the_full_batch(env):
    """configs[4] style data parallelism: a rank that runs all draws on ITS images produces exactly the rows a single
    device computes for them (weight noise does not depend on the batch), incl. the 224x224 flatten quirk."""
    ens = env["ens"]
    net = build(env, "alexnet", "bbb", 10, seed=4)
    x = torch.rand(16, 3, 224, 224, device="cuda")
    E = 2
    with torch.no_grad():
        env["rng"].manual_seed(11, call=0)
        full, kl = ens.mc_forward(net, x, E)
        parts = []
        for r in range(2):
            env["rng"].manual_seed(11, call=0)
            lo, kl_r = ens.mc_forward_batch_parallel(net, x[r * 8:(r + 1) * 8], E)
            parts.append(lo)
            assert kl_r.item() == kl.item()
    assert torch.equal(torch.cat(parts), full)


@pytest.mark.parametrize("hw,B", [(32, 64), (224, 16)])
def test_batch_parallel_lrt_shards_draw_the_full_batchs_noise(env, hw, B):
    """Local-reparameterisation layers draw noise per ACTIVATION, keyed by the GLOBAL image index: a shard that is told where
    its images sit in the full batch (b_offset; default rank * B_local) reproduces the full batch's rows bit for bit -- also
    through the 224x224 flatten quirk, where one image becomes 49 rows of the classifier's input."""
    ens = env["ens"]
    net = build(env, "alexnet", "lrt", 10, seed=5)
    x = torch.rand(B, 3, hw, hw, device="cuda")
    E, half = 2, B // 2
    with torch.no_grad():
        env["rng"].manual_seed(13, call=0)
        full, kl = ens.mc_forward(net, x, E)
        rows = full.shape[0] // 2
        for r in range(2):
            env["rng"].manual_seed(13, call=0)
            lo, kl_r = ens.mc_forward_batch_parallel(net, x[r * half:(r + 1) * half], E, b_offset=r * half)
            assert kl_r.item() == kl.item()
            assert torch.equal(lo, full[r * rows:(r + 1) * rows]), r
        env["rng"].manual_seed(13, call=0)
        wrong, _ = ens.mc_forward_batch_parallel(net, x[half:], E, b_offset=0)         # without the offset: other noise
        assert not torch.equal(wrong, full[rows:])


class _Block(nn.Sequential):
    pass


def test_nested_sequential_model_is_flattened(env):
    """A Bayesian layer inside nn.Sequential (which the reference's ModuleWrapper.forward supports): the batched ensemble path
    must sample E weight sets for it too -- batched == Python loop of net(x), bitwise."""
    L, ens = env["layers"], env["ens"]
    torch.manual_seed(0)

    class Net(L.ModuleWrapper):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(L.BBB_Conv2d(3, 8, 3, padding=1, priors=P.CONFIG_PRIORS), nn.Softplus(), nn.MaxPool2d(2, 2),
                                          _Block(L.BBB_Conv2d(8, 16, 3, padding=1, priors=P.CONFIG_PRIORS), nn.Softplus()))
            self.flatten = L.FlattenLayer(16 * 4 * 4)
            self.fc = L.BBB_Linear(16 * 4 * 4, 10, priors=P.CONFIG_PRIORS)
            self.num_classes = 10

    net = Net().cuda()
    env["rng"].assign_stream_ids(net)
    assert len(ens.flat_children(net)) == 7
    x = torch.rand(8, 3, 8, 8, device="cuda")
    with torch.no_grad():
        nchw, _ = ens.mc_logits(net, x, 3, 5, 0, fuse_act=False, layout="nchw")
        fast, _ = ens.mc_logits(net, x, 3, 5, 0)
        assert ens.stats["path"] == "chwn"
        from layers.misc import reference_layout
        with reference_layout():
            env["rng"].manual_seed(5, call=0)
            loop = torch.stack([net(x)[0] for _ in range(3)])
        env["rng"].manual_seed(5, call=0)
        fast_loop = torch.stack([net(x)[0] for _ in range(3)])
    assert torch.equal(nchw, loop) and torch.equal(fast, fast_loop)
    assert not torch.equal(loop[0], loop[1])                       # three different weight draws
    np.testing.assert_allclose(fast.cpu().numpy(), loop.cpu().numpy(), rtol=5e-4, atol=1e-5)


def test_opaque_container_falls_back_to_the_reference_loop(env):
    """A Bayesian layer hidden inside a module with its own forward cannot be batched: mc_logits must run the reference's
    loop (E different draws), never fold the draws into the batch of ONE weight sample."""
    L, ens = env["layers"], env["ens"]
    torch.manual_seed(0)

    class Hidden(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = L.BBB_Linear(12, 12, priors=P.CONFIG_PRIORS)

        def forward(self, x):
            return x + self.lin(x)

    class Net(L.ModuleWrapper):
        def __init__(self):
            super().__init__()
            self.fc1 = L.BBB_Linear(6, 12, priors=P.CONFIG_PRIORS)
            self.res = Hidden()
            self.fc2 = L.BBB_Linear(12, 4, priors=P.CONFIG_PRIORS)
            self.num_classes = 4

    net = Net().cuda()
    env["rng"].assign_stream_ids(net)
    assert ens.flat_children(net) is None
    x = torch.rand(8, 6, device="cuda")
    with torch.no_grad():
        got, kl = ens.mc_logits(net, x, 4, 21, 3)
        assert ens.stats["path"] == "loop"
        env["rng"].manual_seed(21, call=3)
        loop = torch.stack([net(x)[0] for _ in range(4)])
        env["rng"].manual_seed(21, call=3)
        lo, klsum = ens.mc_forward(net, x, 4)
    assert torch.equal(got, loop) and not torch.equal(got[0], got[1])
    want = torch.logsumexp(torch.log_softmax(loop, dim=2), dim=0) - math.log(4)
    np.testing.assert_allclose(lo.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)
