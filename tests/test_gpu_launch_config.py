"""ops.LaunchConfig: the arithmetic-mode / launch-shape choices are a per-pipeline object, not process-wide switches -- two
pipelines with different modes coexist, interleaved on their own streams, and each computes exactly what it computes alone.
Run with -m gpu."""
import threading

import pytest
import torch

import ref_port_torch as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import layers  # noqa: F401
    from bbb_hip import ops, rng, ensemble, zoo
    return dict(ops=ops, rng=rng, ens=ensemble, zoo=zoo)


def _net(env, B=512):
    torch.manual_seed(3)
    net = env["zoo"].getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
    env["rng"].assign_stream_ids(net)
    return net, torch.rand(B, 3, 32, 32, device="cuda")


def _alone(env, net, x, E, mode, steps, **kw):
    """`steps` results of a pipeline built and run alone under `mode`, from noise call 0 of seed 11."""
    ops, rng, ens = env["ops"], env["rng"], env["ens"]
    rng.manual_seed(11)
    with torch.no_grad(), ops.use_config(gemm_mode=mode):
        pipe = ens.GraphedPipeline(net, x, E, **kw)
    assert ops.current_config().gemm_mode == "fp32"
    out = []
    with torch.no_grad():
        for _ in range(steps):
            lo, kl = pipe.step()
            pipe.sync()
            out.append((lo.clone(), kl.clone()))
    return out


def test_an_fp32_and_a_split_bf16_pipeline_interleaved_on_two_streams(env):
    ops, rng, ens = env["ops"], env["rng"], env["ens"]
    net, x = _net(env)
    E, STEPS = 10, 4
    want32 = _alone(env, net, x, E, "fp32", STEPS, depth=2)
    want16 = _alone(env, net, x, E, "bf16x3", STEPS, depth=2)
    d = max((a[0] - b[0]).abs().max().item() for a, b in zip(want32, want16))
    assert 0 < d < 1e-3, d                                   # the two modes really are different arithmetic (rounding-level)
    # both alive at once, replayed alternately without a sync in between; each lane pool stream carries both pipelines' graphs
    rng.manual_seed(11)
    with torch.no_grad(), ops.use_config(gemm_mode="fp32"):
        p32 = ens.GraphedPipeline(net, x, E, depth=2)
    rng.manual_seed(11)
    with torch.no_grad():
        p16 = ens.GraphedPipeline(net, x, E, depth=2, launch_config=ops.current_config().copy(gemm_mode="bf16x3"))
    assert p32.launch_config.gemm_mode == "fp32" and p16.launch_config.gemm_mode == "bf16x3"
    assert all(l.launch_config.launches_overlap for l in p32.lanes)          # depth 2: the lanes know they overlap
    ops.gemm_mode = "bf16x3"                                 # a later change of the process default reaches neither pipeline
    try:
        got32, got16 = [], []
        with torch.no_grad():
            for _ in range(STEPS):
                a = p32.step()
                b = p16.step()
                torch.cuda.synchronize()
                got32.append((a[0].clone(), a[1].clone()))
                got16.append((b[0].clone(), b[1].clone()))
    finally:
        ops.gemm_mode = "fp32"
    for i in range(STEPS):
        assert torch.equal(got32[i][0], want32[i][0]) and torch.equal(got32[i][1], want32[i][1]), f"fp32 step {i}"
        assert torch.equal(got16[i][0], want16[i][0]) and torch.equal(got16[i][1], want16[i][1]), f"bf16x3 step {i}"


def test_two_threads_with_different_modes(env):
    """use_config is thread-local: a thread that runs eager split-bf16 steps does not disturb another thread's fp32 steps."""
    ops, rng, ens = env["ops"], env["rng"], env["ens"]
    net, x = _net(env, B=256)
    E = 10
    with torch.no_grad():
        rng.manual_seed(5)
        seed, call0 = rng.next_calls(E)
        want32 = ens._local_lse(net, x, E, seed, call0, E)[0].clone()
        with ops.use_config(gemm_mode="bf16x3"):
            want16 = ens._local_lse(net, x, E, seed, call0, E)[0].clone()
    assert not torch.equal(want32, want16)
    res, barrier = {}, threading.Barrier(2)

    def worker(tag, mode, stream):
        with torch.no_grad(), torch.cuda.stream(stream), ops.use_config(gemm_mode=mode):
            barrier.wait()
            outs = [ens._local_lse(net, x, E, seed, call0, E)[0].clone() for _ in range(6)]
            stream.synchronize()
            res[tag] = outs

    ts = [threading.Thread(target=worker, args=("a", "fp32", torch.cuda.Stream())),
          threading.Thread(target=worker, args=("b", "bf16x3", torch.cuda.Stream()))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(torch.equal(o, want32) for o in res["a"]) and all(torch.equal(o, want16) for o in res["b"])
    assert ops.current_config() is ops._default_config and ops.gemm_mode == "fp32"
