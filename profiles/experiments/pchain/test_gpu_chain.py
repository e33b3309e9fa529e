"""The persistent chain launch (bbb_chain_fwd, csrc/pchain.hip: every conv / linear / pool stage of a Monte-Carlo step in ONE
launch, ready-first scheduling with per-(stage, draw) completion counters) against one launch per layer: the GEMM items run
the same instruction sequence, so the comparison is BITWISE, and the scheduler's error word must stay 0."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("unsplit")]   # bitwise comparisons across launch sizes

PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}


def _net(kind, classes, B):
    from bbb_hip import rng, zoo
    torch.manual_seed(0)
    net = zoo.getModel(kind, 3, classes, PRI, "bbb", "softplus").cuda()
    rng.assign_stream_ids(net)
    return net, torch.rand(B, 3, 32, 32, device="cuda")


def _logits(net, x, E, chain, flags=0, units=None):
    from bbb_hip import ensemble
    saved = ensemble.use_chain, ensemble.chain_flags
    ensemble.use_chain, ensemble.chain_flags = chain, flags
    try:
        with torch.no_grad():
            out = ensemble._mc_logits_chwn(net, x, E, 7, 3, units=units)
        torch.cuda.synchronize()
        return out[0].clone(), out[1].clone(), ensemble.stats["launch"]
    finally:
        ensemble.use_chain, ensemble.chain_flags = saved


@pytest.mark.parametrize("kind,classes,B,E", [("alexnet", 10, 512, 10), ("alexnet", 10, 512, 1), ("alexnet", 100, 256, 3),
                                              ("3conv3fc", 10, 256, 2), ("alexnet", 10, 128, 25)])
def test_chain_equals_per_layer_launches_bitwise(kind, classes, B, E):
    from bbb_hip import ops
    net, x = _net(kind, classes, B)
    a, kla, la = _logits(net, x, E, False)
    assert la == "layers"
    for flags in (0, 1):                       # deepest-ready-first, shallowest-ready-first
        b, klb, lb = _logits(net, x, E, True, flags)
        assert lb == "chain"
        assert torch.equal(a, b) and torch.equal(kla, klb)
        assert ops.chain_error(x.device) == 0


def test_chain_work_units_bitwise():
    """A rank's share of the strong-scaling step (8 ranks: five quarter-batch units) through the chain."""
    from bbb_hip import ensemble, ops
    net, x = _net("alexnet", 10, 512)
    S = ensemble.plan_slices(10, 8, 512)
    for rank in (0, 3, 7):
        lo, hi = ensemble.unit_range(10, S, rank, 8)
        a, _, _ = _logits(net, x, 10, False, units=(S, lo, hi))
        b, _, lb = _logits(net, x, 10, True, units=(S, lo, hi))
        assert lb == "chain" and torch.equal(a, b) and ops.chain_error(x.device) == 0


def test_chain_in_graph_lanes_bitwise():
    """Three steps in flight as hipGraph lanes (the benchmark's launch mode), chain vs per-layer, every step compared."""
    from bbb_hip import ensemble, ops, rng
    net, x = _net("alexnet", 10, 512)
    outs = {}
    saved = ensemble.use_chain
    try:
        for chain in (False, True):
            ensemble.use_chain = chain
            rng.manual_seed(11, 0)
            with torch.no_grad():
                pipe = ensemble.GraphedPipeline(net, x, 10, depth=3)
                res = []
                for _ in range(12):
                    lo, _ = pipe.step()
                res_sync = None
                pipe.sync()
                for _ in range(6):
                    lo, _ = pipe.step()
                    pipe.sync()
                    res.append(lo.clone())
            outs[chain] = res
            del pipe
    finally:
        ensemble.use_chain = saved
    assert all(torch.equal(u, v) for u, v in zip(outs[False], outs[True]))
    assert all(int(buf[8].item()) == 0 for k, buf in ops._scratch.items() if k[1] == "chain")


def test_chain_falls_back_where_it_does_not_apply():
    """LRT layers, B % 128 != 0 and more than 64 draws keep the per-layer launches (same results either way)."""
    from bbb_hip import ensemble, rng, zoo
    net, x = _net("alexnet", 10, 64)
    assert _logits(net, x, 2, True)[2] == "layers"
    net, x = _net("alexnet", 10, 128)
    a, _, la = _logits(net, x, 70, False)
    b, _, lb = _logits(net, x, 70, True)
    assert lb == "layers" and torch.equal(a, b)
    torch.manual_seed(0)
    lrt = zoo.getModel("alexnet", 3, 10, PRI, "lrt", "softplus").cuda()
    rng.assign_stream_ids(lrt)
    assert _logits(lrt, torch.rand(128, 3, 32, 32, device="cuda"), 2, True)[2] == "layers"
