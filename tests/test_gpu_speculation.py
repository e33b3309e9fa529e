"""Speculative draw batching behind `net(x)` (layers/_fused.py): the reference's Monte-Carlo loop calls net(inputs) num_ens
times on the SAME tensor (main_bayesian.py:43-53, :73-80); repeated calls are answered from ONE batched launch whose draw j
uses the call index the j-th call of the loop would have used.  By the noise contract that is bit-identical to the loop, and
the generator offset must advance exactly as the loop advances it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu          # bitwise comparisons across launch sizes, at the SHIPPED defaults (ops.split_k on)
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}


def _net(lt="bbb", B=128, grad=False):
    from bbb_hip import rng, zoo
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, 10, PRI, lt, "softplus").cuda()
    rng.assign_stream_ids(net)
    for p in net.parameters():
        p.requires_grad_(grad)
    return net, torch.rand(B, 3, 32, 32, device="cuda")


def _loops(net, xs, E, spec, seed=5, mutate=None):
    """len(xs) validation-style loops of E calls; returns every (output, kl) and the generator offset afterwards."""
    from layers import _fused
    saved = _fused.speculation["enabled"]
    _fused.speculation["enabled"] = spec
    net.__dict__.pop("_bbb_spec", None)
    torch.manual_seed(seed)
    outs = []
    try:
        for li, x in enumerate(xs):
            for j in range(E):
                if mutate is not None:
                    mutate(li, j, x, net)
                y, kl = net(x)
                outs.append((y.detach().clone(), kl.detach().clone()))
    finally:
        _fused.speculation["enabled"] = saved
    return outs, torch.cuda.default_generators[torch.cuda.current_device()].get_offset()


def _same(a, b):
    return len(a) == len(b) and all(torch.equal(u[0], v[0]) and torch.equal(u[1], v[1]) for u, v in zip(a, b))


def test_speculated_loop_is_the_loop_bit_for_bit():
    net, x = _net()
    xs = [x, torch.rand_like(x), torch.rand_like(x)]
    with torch.no_grad():
        ref, off_ref = _loops(net, xs, 10, False)
        got, off_got = _loops(net, xs, 10, True)
    assert _same(ref, got) and off_ref == off_got
    sp = net.__dict__["_bbb_spec"]
    assert sp.last_streak == 10 and sp.streak == 10          # the third loop ran as ONE batched launch of 10 draws


def test_changes_in_mid_loop_give_the_loops_bits():
    net, x = _net()
    xs = [x, torch.rand_like(x), torch.rand_like(x), torch.rand_like(x)]

    def mutate(li, j, xx, nn_):
        if li == 1 and j == 4:
            xx.mul_(0.5)                                     # the input changes in place between two calls
        if li == 2 and j == 6:
            with torch.no_grad():
                next(nn_.parameters()).add_(0.01)            # a parameter moves (version counter)
        if li == 3 and j == 3:
            torch.rand(7, device="cuda")                     # somebody else draws from the generator
    with torch.no_grad():
        net_a, _ = _net()
        ref, off_ref = _loops(net_a, [t.clone() for t in xs], 10, False, mutate=mutate)
        net_b, _ = _net()
        got, off_got = _loops(net_b, [t.clone() for t in xs], 10, True, mutate=mutate)
    assert _same(ref, got) and off_ref == off_got


def test_a_new_tensor_at_the_old_address_is_a_new_input():
    from layers import _fused
    net, x = _net()
    with torch.no_grad():
        torch.manual_seed(3)
        for _ in range(2):
            for _ in range(6):
                net(x)
        y_next = net(x)[0]                                   # 13th call: a new streak starts -> 6 draws speculated
        ptr = x.data_ptr()
        del x
        x2 = torch.rand(128, 3, 32, 32, device="cuda")       # the allocator hands the freed block to the next batch
        got = net(x2)[0].clone()
        off = torch.cuda.default_generators[0].get_offset()
        saved = _fused.speculation["enabled"]
        _fused.speculation["enabled"] = False
        try:
            torch.cuda.default_generators[0].set_offset(off - 4)
            want = net(x2)[0].clone()
        finally:
            _fused.speculation["enabled"] = saved
    assert torch.equal(got, want), ("stale speculative draw served", ptr == x2.data_ptr())


@pytest.mark.parametrize("lt", ["bbb", "lrt"])
def test_speculation_with_autograd_enabled_matches_the_loop_and_its_gradients(lt):
    """validate_model does not disable autograd, train_model backpropagates through all num_ens forwards: the speculated draws
    come from ONE autograd node, so outputs are the loop's bits and the gradients the loop's up to summation order."""
    E = 4

    def run(spec):
        from layers import _fused
        net, x = _net(lt, 64, grad=True)
        y = torch.randint(0, 10, (64,), device="cuda")
        saved = _fused.speculation["enabled"]
        _fused.speculation["enabled"] = spec
        torch.manual_seed(9)
        try:
            outs = []
            for it in range(3):                              # the first iterations teach the streak length
                net.zero_grad()
                loss = 0.0
                for _ in range(E):
                    o, kl = net(x)
                    outs.append(o.detach().clone())
                    loss = loss + F.cross_entropy(o, y) + 1e-6 * kl
                loss.backward()
            grads = [p.grad.clone() for p in net.parameters()]
        finally:
            _fused.speculation["enabled"] = saved
        return outs, grads
    o1, g1 = run(False)
    o2, g2 = run(True)
    assert all(torch.equal(a, b) for a, b in zip(o1, o2))
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-6 * float(a.abs().max()) + 1e-12)


def test_hooks_and_input_gradients_take_the_reference_layout_path():
    """The fused whole-model forward does not call the children and has no d/dx: a registered forward hook, or an input that
    requires a gradient, sends net(x) down the per-layer path."""
    net, x = _net(grad=True)
    fired = []
    h = net.conv1.register_forward_hook(lambda m, i, o: fired.append(tuple(o.shape)))
    net(x)
    h.remove()
    assert fired and fired[0][1] == 64
    xg = x.clone().requires_grad_(True)
    out, kl = net(xg)
    out.sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0
    for p in net.parameters():
        p.requires_grad_(False)
    xg2 = x.clone().requires_grad_(True)                     # frozen parameters, gradient w.r.t. the input only (saliency maps)
    out2, _ = net(xg2)
    out2.sum().backward()
    assert xg2.grad is not None and float(xg2.grad.abs().max()) > 0


def test_speculated_batches_served_from_a_cached_graph_are_the_loop_bit_for_bit():
    """From the second batch of a shape on, a K-draw batch under no_grad is ONE hipGraph replay (ensemble.GraphedLogits): same
    bits as the loop of single calls, same generator offsets, results owned by the caller (not the graph's buffers), a
    parameter update between batches is seen (the graph reads the live parameters), new parameter storage drops the graph."""
    from layers import _fused
    from bbb_hip import ensemble
    net, x = _net()
    xs = [x] + [torch.rand_like(x) for _ in range(5)]
    with torch.no_grad():
        ref, off_ref = _loops(net, xs, 10, False)
        net.__dict__.pop("_bbb_structure", None)
        got, off_got = _loops(net, xs, 10, True)
    assert _same(ref, got) and off_ref == off_got
    graphs = [e[1] for e in ensemble._structure(net)["logit_graphs"].values() if e[1]]
    assert len(graphs) == 1 and graphs[0].K == 10                       # batches 3..6 were replays of one graph
    # outputs handed out earlier are the caller's: later replays did not overwrite them
    assert torch.equal(got[25][0], ref[25][0]) and got[25][0].is_contiguous()
    # a parameter moves between batches (optimizer step): the replay reads the live value
    def mutate(li, j, xx, nn_):
        if li == 3 and j == 0:
            with torch.no_grad():
                nn_.conv1.W_mu.add_(0.01)
    with torch.no_grad():
        net_a, _ = _net()
        ref2, _ = _loops(net_a, [t.clone() for t in xs], 10, False, mutate=mutate)
        net_b, _ = _net()
        got2, _ = _loops(net_b, [t.clone() for t in xs], 10, True, mutate=mutate)
    assert _same(ref2, got2)
    # new storage for a parameter: a different key -> the old graph is not used for it
    with torch.no_grad():
        net_b.conv1.W_mu.data = net_b.conv1.W_mu.data.clone()
        net_a.conv1.W_mu.data = net_a.conv1.W_mu.data.clone()
        r3, _ = _loops(net_a, [xs[0].clone(), xs[1].clone()], 10, False, seed=9)
        g3, _ = _loops(net_b, [xs[0].clone(), xs[1].clone()], 10, True, seed=9)
    assert _same(r3, g3)
    saved = _fused.speculation["graph_after"]
    _fused.speculation["graph_after"] = 0                                # switch: speculation without graphs
    try:
        with torch.no_grad():
            net_c, _ = _net()
            got4, off4 = _loops(net_c, xs, 10, True)
        assert _same(ref, got4) and off4 == off_ref and "logit_graphs" not in ensemble._structure(net_c)
    finally:
        _fused.speculation["graph_after"] = saved
