"""The noise follows torch's HIP generator ("seed + offset from torch's HIP generator"): seeding, state
save / restore and interleaving with torch's own random kernels govern the on-chip Philox stream; rng.manual_seed pins a
private stream and leaves the generator alone.  Run with -m gpu."""
import pytest
import torch

import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import rng, ensemble, zoo
    torch.manual_seed(0)
    net = zoo.getModel("lenet", 1, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    rng.assign_stream_ids(net)
    return dict(rng=rng, ens=ensemble, net=net, x=torch.rand(8, 1, 32, 32).cuda())


def fwd(env, E=3):
    with torch.no_grad():
        lo, _ = env["ens"].mc_forward(env["net"], env["x"], E)
    return lo.clone()


def test_cuda_manual_seed_governs_the_noise(env):
    torch.cuda.manual_seed(5)
    a = fwd(env)
    b = fwd(env)                                   # the offset moved on: fresh noise
    torch.cuda.manual_seed(5)
    c = fwd(env)
    torch.cuda.manual_seed(6)
    d = fwd(env)
    assert torch.equal(a, c) and not torch.equal(a, b) and not torch.equal(a, d)


def test_generator_state_save_restore_replays_the_noise(env):
    torch.cuda.manual_seed(11)
    fwd(env)
    st = torch.cuda.get_rng_state()
    a = fwd(env)
    torch.cuda.set_rng_state(st)
    b = fwd(env)
    assert torch.equal(a, b)


def test_offsets_interleave_with_torch_kernels_and_advance_by_4_per_call(env):
    rng = env["rng"]
    torch.cuda.manual_seed(3)
    g = torch.cuda.default_generators[torch.cuda.current_device()]
    off0 = g.get_offset()
    assert rng.get_state() == (g.initial_seed() & 0xFFFFFFFFFFFFFFFF, off0 // 4)
    a = fwd(env, E=3)
    assert g.get_offset() == off0 + 4 * 3          # three call indices = three Philox counters' worth of offset
    torch.cuda.manual_seed(3)
    torch.rand(16, device="cuda")                  # a torch kernel consumes offsets first ...
    assert g.get_offset() > off0
    b = fwd(env, E=3)
    assert not torch.equal(a, b)                   # ... so the forward sees later call indices
    with torch.no_grad():
        torch.cuda.manual_seed(3)
        y1, _ = env["net"](env["x"])               # the drop-in forward draws from the same stream: one call index
        assert g.get_offset() == off0 + 4
        torch.cuda.manual_seed(3)
        logits, _ = env["ens"].mc_logits(env["net"], env["x"], 1, *rng.get_state())
    assert torch.equal(y1, logits[0])


def test_pinned_stream_leaves_the_generator_alone(env):
    rng = env["rng"]
    torch.cuda.manual_seed(9)
    g = torch.cuda.default_generators[torch.cuda.current_device()]
    off = g.get_offset()
    rng.manual_seed(1234, call=7)
    a = fwd(env)
    assert g.get_offset() == off and rng.get_state() == (1234, 10)
    rng.manual_seed(1234, call=7)
    assert torch.equal(a, fwd(env))
    torch.manual_seed(1)                           # reseeding torch un-pins
    fwd(env)
    assert g.get_offset() == 4 * 3


def test_graphed_step_keeps_the_host_offset_in_step(env):
    torch.cuda.manual_seed(21)
    g = torch.cuda.default_generators[torch.cuda.current_device()]
    with torch.no_grad():
        gm = env["ens"].GraphedMC(env["net"], env["x"], 4)
        off = g.get_offset()
        a = gm.step()[0].clone()
        b = gm.step()[0].clone()
        torch.cuda.synchronize()
    assert g.get_offset() == off + 2 * 4 * 4 and not torch.equal(a, b)
    torch.cuda.manual_seed(21)
    with torch.no_grad():
        want = fwd(env, E=4)                       # eager step under the same seed = first replay
    assert torch.equal(a, want)


def test_device_noise_distribution_at_1e8_samples():
    """The noise contract's output as a DISTRIBUTION (review r05: one moment test was all that guarded it): 1e8 device normals
    (Philox4x32-7 -> 23-bit uniforms -> Box-Muller with the hardware log / sin / cos; ten (call, stream) pairs of 1e7 elements) --
    mean, variance, skewness, excess kurtosis within 5 standard errors of N(0, 1)'s; tail counts beyond 2, 3 and 4 sigma within 5
    binomial standard deviations; Kolmogorov-Smirnov distance of every 1e7-sample block against the normal CDF below the
    alpha = 0.001 critical value 1.95 / sqrt(n); no value beyond the contract's bound sqrt(-2 ln 2^-23) = 5.65."""
    import math
    from bbb_hip import ops
    n_blk, blocks = 10_000_000, 10
    n = n_blk * blocks
    s1 = s2 = s3 = s4 = 0.0
    tails = {2.0: 0, 3.0: 0, 4.0: 0}
    worst_ks, worst_abs = 0.0, 0.0
    grid = (torch.arange(n_blk, device="cuda", dtype=torch.float64) + 0.5) / n_blk
    for blk in range(blocks):
        z = ops.eps_dump(n_blk, 987654321 + blk, 1000 * blk + 7, blk % 3, torch.device("cuda"))
        zd = z.double()
        s1 += float(zd.sum()); s2 += float((zd * zd).sum()); s3 += float((zd ** 3).sum()); s4 += float((zd ** 4).sum())
        a = z.abs()
        worst_abs = max(worst_abs, float(a.max()))
        for t in tails:
            tails[t] += int((a > t).sum())
        zs, _ = torch.sort(zd)
        cdf = 0.5 * (1.0 + torch.erf(zs / math.sqrt(2.0)))
        worst_ks = max(worst_ks, float((cdf - grid).abs().max()) + 0.5 / n_blk)
        del z, zd, a, zs, cdf
    mean = s1 / n
    var = s2 / n - mean * mean
    skew = (s3 / n - 3 * mean * s2 / n + 2 * mean ** 3) / var ** 1.5
    kurt = (s4 / n - 4 * mean * s3 / n + 6 * mean * mean * s2 / n - 3 * mean ** 4) / (var * var) - 3.0
    print(f"[noise 1e8] mean {mean:.2e} var-1 {var - 1:.2e} skew {skew:.2e} excess kurtosis {kurt:.2e} KS {worst_ks:.2e} max|z| {worst_abs:.3f} "
          f"tails {tails}")
    assert abs(mean) <= 5 / math.sqrt(n) and abs(var - 1) <= 5 * math.sqrt(2 / n)
    assert abs(skew) <= 5 * math.sqrt(6 / n) and abs(kurt) <= 5 * math.sqrt(24 / n)
    for t, cnt in tails.items():
        pr = math.erfc(t / math.sqrt(2.0))
        assert abs(cnt - n * pr) <= 5 * math.sqrt(n * pr * (1 - pr)), (t, cnt, n * pr)
    assert worst_ks <= 1.95 / math.sqrt(n_blk)
    assert worst_abs <= 5.66
