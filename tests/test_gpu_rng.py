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
