"""Batches that are not a multiple of 4 stay on the batch-innermost kernels (review r04 item 9): zero images pad the batch, their
rows are dropped.  The real images' results are bit for bit those of the same images inside a full batch (every image is its own
GEMM column; LRT noise is keyed by the global image index), and agree with the reference-layout path to rounding.  Run with -m gpu."""
import pytest
import torch

import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo)


@pytest.mark.parametrize("name,cin,lt,B", [("alexnet", 3, "bbb", 510), ("alexnet", 3, "lrt", 510), ("lenet", 1, "bbb", 7),
                                           ("3conv3fc", 3, "lrt", 1), ("lenet", 1, "lrt", 62)])
def test_odd_batch_is_the_padded_batch_without_its_padding(env, name, cin, lt, B):
    ops, rng, ens, zoo = env["ops"], env["rng"], env["ens"], env["zoo"]
    torch.manual_seed(B)
    net = zoo.getModel(name, cin, 10, P.CONFIG_PRIORS, lt, "softplus").cuda()
    rng.assign_stream_ids(net)
    Bp = -(-B // 4) * 4
    xfull = torch.rand(Bp, cin, 32, 32, device="cuda")
    x = xfull[:B].clone()
    E, seed, call0 = 3, 77, 9
    with torch.no_grad():
        got, kl = ens.mc_logits(net, x, E, seed, call0)                       # [E, B, C]
        assert ens.stats["path"] == "chwn", ens.stats                       # the fast path took it
        full, kl_full = ens.mc_logits(net, xfull, E, seed, call0)
        ref, kl_ref = ens.mc_logits(net, x, E, seed, call0, layout="nchw")
        assert ens.stats["path"] == "nchw"
        lo, _ = ens._local_lse(net, x, E, seed, call0, E)
        lo_full, _ = ens._local_lse(net, xfull, E, seed, call0, E)
    assert got.shape == (E, B, 10) and lo.shape == (B, 10)
    assert torch.equal(got, full[:, :B]) and torch.equal(kl, kl_full)
    assert torch.equal(lo, lo_full[:B])
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, scale)           # the two layouts' bound of the full-size tests
    assert abs(kl.item() - kl_ref.item()) <= 2e-6 * abs(kl_ref.item())


def test_dropin_forward_and_graphed_step_on_an_odd_batch(env):
    ops, rng, ens, zoo = env["ops"], env["rng"], env["ens"], env["zoo"]
    torch.manual_seed(1)
    net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    rng.assign_stream_ids(net)
    x = torch.rand(510, 3, 32, 32, device="cuda")
    with torch.no_grad():
        rng.manual_seed(3)
        out, kl = net(x)                                                      # the drop-in call
        assert out.shape == (510, 10) and torch.isfinite(out).all()
        rng.manual_seed(3)
        seed, call0 = rng.next_calls(1)
        want, _ = ens.mc_logits(net, x, 1, seed, call0)
        assert torch.equal(out, want[0])
        rng.manual_seed(5)
        g = ens.GraphedMC(net, x, 10)
        lo, klg = g.step()
        torch.cuda.synchronize()
        lo = lo.clone()
        rng.manual_seed(5)
        seed, call0 = rng.next_calls(10)
        lo_e, _ = ens._local_lse(net, x, 10, seed, call0, 10)
    assert lo.shape == (510, 10) and torch.equal(lo, lo_e)
