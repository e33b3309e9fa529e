"""Several Monte-Carlo steps per launch (GraphedPipeline steps_per_launch, GraphedMC steps; one-draw steps since round 3, any
num_ens since round 4: G steps x E draws = G * E slabs per launch, bbb_conv_desc_t::x_unit_div + bbb_mc_tail_groups_step): every
step must be what the one-step-per-replay pipeline computes for the same batch and the same noise calls -- bit for bit on the
fp32 BBB / LRT kernels, to bf16 storage rounding on the bf16 path.  Run with -m gpu."""
import pytest
import torch

import ref_port_torch as P

pytestmark = pytest.mark.gpu          # bitwise comparisons across launch sizes, at the SHIPPED defaults (ops.split_k on)


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ensemble, rng, zoo
    return dict(ens=ensemble, rng=rng, zoo=zoo)


def _net(env, kind, layer_type, classes=10):
    torch.manual_seed(4)
    net = env["zoo"].getModel(kind, 3, classes, P.CONFIG_PRIORS, layer_type, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    return net


def _run(env, net, batches, G, depth, precision, E=1):
    env["rng"].manual_seed(21, call=0)
    with torch.no_grad():
        pipe = env["ens"].GraphedPipeline(net, batches[0], E, depth=depth, precision=precision, steps_per_launch=G)
        outs, views = [], []
        for i, xb in enumerate(batches):
            views.append(pipe.step(xb))
            if G == 1 or (i + 1) % (depth * G) == 0 or i + 1 == len(batches):
                pipe.sync()                                              # (flushes a partly filled group)
                outs += [(lo.clone(), kl.clone()) for lo, kl in views]
                views = []
    return outs, env["rng"].get_state()[1]


@pytest.mark.parametrize("kind,layer_type,precision,B", [("alexnet", "bbb", "fp32", 128), ("alexnet", "lrt", "fp32", 64),
                                                         ("3conv3fc", "bbb", "fp32", 32), ("3conv3fc", "bbb", "bf16", 64)])
def test_grouped_steps_equal_single_steps(env, kind, layer_type, precision, B):
    net = _net(env, kind, layer_type)
    torch.manual_seed(9)
    batches = [torch.rand(B, 3, 32, 32, device="cuda") for _ in range(11)]
    ref, calls1 = _run(env, net, batches, 1, 3, precision)
    got, calls4 = _run(env, net, batches, 4, 3, precision)
    assert calls1 == 11 and calls4 == 12                                 # the flushed group consumed its empty slot's call too
    for i, ((lo_r, kl_r), (lo_g, kl_g)) in enumerate(zip(ref, got)):
        assert lo_g.shape == lo_r.shape == (B, 10)
        assert torch.equal(kl_g, kl_r)
        if precision == "fp32":
            assert torch.equal(lo_g, lo_r), f"step {i}: max diff {float((lo_g - lo_r).abs().max()):.3e}"
        else:       # bf16: the in-workgroup k-group choice depends on the launch size -> fp32 summation order -> bf16 rounding flips
            assert float((lo_g - lo_r).abs().max()) <= 2e-2 * float(lo_r.abs().max())
    # two different batches in one group really got different weights AND different inputs
    assert not torch.equal(got[0][0], got[1][0])


@pytest.mark.parametrize("kind,layer_type,precision,B,E,G", [("alexnet", "bbb", "fp32", 128, 3, 2), ("alexnet", "bbb", "fp32", 64, 10, 4),
                                                             ("alexnet", "lrt", "fp32", 64, 2, 3), ("lenet", "bbb", "fp32", 32, 4, 2),
                                                             ("3conv3fc", "bbb", "bf16", 64, 2, 2)])
def test_grouped_multi_draw_steps_equal_single_steps(env, kind, layer_type, precision, B, E, G):
    """G steps x E draws per launch: slab g * E + j = draw j of step g (call g * E + j) on batch g; every step's log_outputs and
    KL equal the one-step-per-launch pipeline's, and the host call counter advances E per step (flushed slots included)."""
    torch.manual_seed(4)
    cin = 1 if kind == "lenet" else 3
    net = env["zoo"].getModel(kind, cin, 10, P.CONFIG_PRIORS, layer_type, "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    torch.manual_seed(9)
    n = 2 * 2 * G + 1                                                    # two full rounds of two lanes + one step in a flushed group
    batches = [torch.rand(B, cin, 32, 32, device="cuda") for _ in range(n)]
    ref, calls1 = _run(env, net, batches, 1, 2, precision, E)
    got, callsG = _run(env, net, batches, G, 2, precision, E)
    assert calls1 == n * E and callsG == (n + G - 1) * E
    for i, ((lo_r, kl_r), (lo_g, kl_g)) in enumerate(zip(ref, got)):
        assert lo_g.shape == lo_r.shape == (B, 10)
        assert torch.equal(kl_g, kl_r)
        if precision == "fp32":
            assert torch.equal(lo_g, lo_r), f"step {i}: max diff {float((lo_g - lo_r).abs().max()):.3e}"
        else:
            assert float((lo_g - lo_r).abs().max()) <= 2e-2 * float(lo_r.abs().max())
    assert not torch.equal(got[0][0], got[1][0])


def test_grouped_tail_kernel_against_torch(env):
    """bbb_mc_tail_groups_step on its own: [G * E, C, B] -> block g = logmeanexp over step g's draws of log_softmax."""
    from bbb_hip import ops
    torch.manual_seed(2)
    G, E, C, B = 3, 5, 10, 72
    logits = torch.randn(G * E, C, B, device="cuda") * 3
    kl = torch.tensor(2.5, device="cuda")
    counter = torch.tensor([7], dtype=torch.int32, device="cuda")
    out, klo = ops.mc_tail_groups(logits, G, E, mean_over=E, step_end=(kl, 4.0, counter, 11))
    want = torch.logsumexp(torch.log_softmax(logits.view(G, E, C, B), dim=2), dim=1) - torch.log(torch.tensor(float(E)))
    want = want.permute(0, 2, 1).reshape(G * B, C)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    assert klo.item() == 10.0 and counter.item() == 18
    plain = ops.mc_tail_groups(logits, G, E, mean_over=E)
    assert torch.equal(plain, out)
    for g in range(G):                                                   # block g = the ordinary tail of step g's slabs
        assert torch.equal(out[g * B:(g + 1) * B], ops.mc_tail_cb(logits[g * E:(g + 1) * E], mean_over=E))


def test_grouped_steps_keep_going_after_a_flush(env):
    """sync() in the middle of a group, then more steps: still the single-step sequence (call indices realigned to group starts)."""
    net = _net(env, "alexnet", "bbb")
    torch.manual_seed(10)
    batches = [torch.rand(64, 3, 32, 32, device="cuda") for _ in range(10)]
    env["rng"].manual_seed(33, call=0)
    with torch.no_grad():
        pipe = env["ens"].GraphedPipeline(net, batches[0], 1, depth=2, steps_per_launch=4)
        first = [pipe.step(b) for b in batches[:2]]
        pipe.sync()
        first = [lo.clone() for lo, _ in first]
        rest = [pipe.step(b) for b in batches[2:6]]                      # a full group on the next lane: calls 4..7
        pipe.sync()
        rest = [lo.clone() for lo, _ in rest]
        single = []
        for call, xb in zip([0, 1, 4, 5, 6, 7], batches[:6]):
            env["rng"].manual_seed(33, call=call)
            single.append(env["ens"].mc_forward(net, xb, 1)[0])
    for a, b in zip(first + rest, single):
        assert torch.equal(a, b)


def test_grouped_steps_refuse_what_they_do_not_cover(env):
    from bbb_hip._lib import BBBHipError
    net = _net(env, "alexnet", "bbb")
    x = torch.rand(64, 3, 32, 32, device="cuda")
    with torch.no_grad(), pytest.raises(BBBHipError):
        env["ens"].GraphedPipeline(net, x[:6], 2, depth=2, steps_per_launch=4)      # B % 4 != 0: no batch-innermost path


@pytest.mark.parametrize("lt,E,G,world,precision", [("bbb", 10, 4, 8, "fp32"), ("bbb", 10, 4, 3, "fp32"), ("lrt", 4, 3, 2, "fp32"),
                                                    ("bbb", 1, 4, 2, "fp32"), ("bbb", 3, 2, 8, "fp32"), ("lrt", 5, 2, 4, "fp32"),
                                                    ("bbb", 10, 4, 8, "bf16"), ("bbb", 5, 3, 4, "bf16x3")])
def test_group_of_steps_dealt_to_ranks_is_the_steps(lt, E, G, world, precision):
    """N > 1 with several steps per launch (ensemble.group_share): the G * E draws of a group, draw-major, in `world` contiguous
    ranges -- whole draws on whole batches; a rank's range may start and end in the middle of a step.  Every rank's logits are
    the slabs the single steps compute (fp32 kernels: bitwise); the ranks' blocks combined by one log-sum-exp are the steps' results; the KL
    shares add up.  (All ranks simulated on this device; the collective itself: test_gpu_rccl.py.)"""
    import math
    from bbb_hip import ensemble, rng, zoo
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    rng.assign_stream_ids(net)
    B = 64
    xs = [torch.rand(B, 3, 32, 32, device="cuda") for _ in range(G)]
    seed, call0 = 21, 300
    with torch.no_grad():
        ref_logits = [ensemble._mc_logits_chwn(net, xs[g], E, seed, call0 + g * E, precision=precision)[0] for g in range(G)]
        ref = [ensemble._local_lse(net, xs[g], E, seed, call0 + g * E, E, precision=precision) for g in range(G)]
        blocks = torch.full((world, G * B, 10), -float("inf"), device="cuda")
        kl_sum = 0.0
        covered = 0
        for rank in range(world):
            lo, hi, g_lo, n_gl, off = ensemble.group_share(E, G, rank, world)
            if hi <= lo:
                continue
            covered += hi - lo
            xl = torch.cat(xs[g_lo:g_lo + n_gl])
            lg = ensemble._mc_logits_chwn(net, xl, hi - lo, seed, call0 + lo, share=(E, off), precision=precision)[0]
            for e in range(hi - lo):
                d = lo + e
                want_e = ref_logits[d // E][d % E]
                if precision == "fp32":                      # the fp32 kernels: same bits at every launch size
                    assert torch.equal(lg[e], want_e), (rank, e)
                else:                                        # bf16 / split-bf16 pick their kernel form by launch size: same values
                    tol = (2e-2 if precision == "bf16" else 2e-6) * float(want_e.abs().max())
                    assert torch.allclose(lg[e], want_e, rtol=0, atol=tol), (rank, e)
            lse, kl1 = ensemble._local_lse(net, xl, hi - lo, seed, call0 + lo, 0, share=(E, off), precision=precision)
            assert lse.shape == (n_gl * B, 10)
            blocks[rank, g_lo * B:(g_lo + n_gl) * B] = lse
            kl_sum = kl_sum + kl1 * float(hi - lo)
    assert covered == G * E
    got = torch.logsumexp(blocks, dim=0) - math.log(E)
    want = torch.cat([r[0] for r in ref])
    assert torch.allclose(got, want, rtol=0, atol=(3e-2 if precision == "bf16" else 3e-6) * max(1.0, float(want.abs().max())))
    assert abs(float(kl_sum) / G - E * float(ref[0][1])) <= 1e-6 * abs(E * float(ref[0][1]))
