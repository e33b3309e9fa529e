"""The N > 1 step protocol through RCCL itself (backend "nccl").
(1) On ONE MI355X: a world-size-1 process group with the sharded code path forced (BBB_FORCE_COMBINE=1) -- the step's units,
ONE all_gather_into_tensor and the reduction over ranks, recorded into one hipGraph when RCCL allows capture (probed), else
replay + eager all_gather + replay -- must reproduce the plain single-process step, in both protocols.
(2) On a box with >= 2 GPUs (auto-skipped below): min(device_count, 8) ranks, one process per GPU, the metric's 512 x 10 step
sharded into work units over RCCL: every rank's log_outputs / kl equal the single-device step's (bitwise: the rank-order
log-sum-exp is the same arithmetic on every rank; vs the one-GPU tail to 3e-6).  This is the test that runs RCCL with N > 1 the
day a multi-GPU box runs the suite.  (More than one rank per device is not something RCCL allows; world sizes 2-8 are also
covered with gloo in test_host_cpu.py and with simulated ranks in test_gpu_sharding.py.)
(3) The same worker with 2 and 3 real ranks on ONE device over gloo: both shardings (work units of one step; groups of four steps
dealt out as whole draws) against the single-device step, through the eager collective protocol.
Workers run in subprocesses: a process group is process-global state.  Run with -m gpu."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "pytorch-bayesiancnn_amd")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29547"
import torch, torch.distributed as dist
import ref_port_torch as P
from bbb_hip import ensemble, zoo, rng
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
group = dist.group.WORLD
out = {}
for lt, ncls in (("bbb", 10), ("lrt", 100)):
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, ncls, P.CONFIG_PRIORS, lt, "softplus").to(dev)
    rng.assign_stream_ids(net)
    x = torch.rand(64, 3, 32, 32, device=dev)
    E = 4
    # eager entry point: with and without the group
    rng.manual_seed(7, 100)
    with torch.no_grad():
        lo0, kl0 = ensemble.mc_forward(net, x, E)
    rng.manual_seed(7, 100)
    with torch.no_grad():
        lo1, kl1 = ensemble.mc_forward(net, x, E, group=group)
    # graph lanes: plain vs the forced multi-rank protocol -- recorded into the step's graph (RCCL capture) and eager
    res = {}
    fused = {}
    for tag, grp, force, cap in (("plain", None, "0", True), ("rccl", group, "1", True), ("rccl_eager", group, "1", False)):
        os.environ["BBB_FORCE_COMBINE"] = force
        ensemble.capture_collectives = cap
        ensemble._capture_probe.clear()
        rng.manual_seed(7, 200)
        pipe = ensemble.GraphedPipeline(net, x, E, depth=2, group=grp)
        fused[tag] = [bool(l.fused) for l in pipe.lanes]
        steps = []
        for _ in range(5):
            lo, kl = pipe.step()
            pipe.sync()
            steps.append((lo.clone(), kl.clone()))
        res[tag] = steps
        del pipe
    # a GROUP of 2 steps per launch dealt to the ranks (group_share), recorded and eager: same steps, same noise calls
    grp_res = {}
    for tag, cap in (("fused", True), ("eager", False)):
        os.environ["BBB_FORCE_COMBINE"] = "1"
        ensemble.capture_collectives = cap
        ensemble._capture_probe.clear()
        rng.manual_seed(7, 200)
        pipe = ensemble.GraphedPipeline(net, x, E, depth=2, group=group, steps_per_launch=2)
        views = [pipe.step() for _ in range(4)]
        pipe.sync()
        grp_res[tag] = [(a.clone(), b.clone()) for a, b in views]
        del pipe
    os.environ["BBB_FORCE_COMBINE"] = "0"
    ensemble.capture_collectives = True
    d_group = max((a[0] - b[0]).abs().max().item() for a, b in zip(res["plain"][:4], grp_res["fused"]))
    d_group_kl = max(abs(a[1].item() - b[1].item()) / abs(a[1].item()) for a, b in zip(res["plain"][:4], grp_res["fused"]))
    group_same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(grp_res["fused"], grp_res["eager"]))
    same_protocols = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(res["rccl"], res["rccl_eager"]))
    d_eager = (lo1 - lo0).abs().max().item()
    d_graph = max((a[0] - b[0]).abs().max().item() for a, b in zip(res["plain"], res["rccl"]))
    d_kl = max(abs(a[1].item() - b[1].item()) / abs(a[1].item()) for a, b in zip(res["plain"], res["rccl"]))
    fresh = not torch.equal(res["rccl"][0][0], res["rccl"][1][0])
    out[lt] = dict(d_eager=d_eager, kl_eager=abs(kl1.item() - kl0.item()) / abs(kl0.item()), d_graph=d_graph, d_kl=d_kl, fresh=fresh,
                   scale=lo0.abs().max().item(), finite=bool(torch.isfinite(res["rccl"][-1][0]).all()),
                   fused=fused, same_protocols=same_protocols, d_group=d_group, d_group_kl=d_group_kl, group_same=group_same)
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_step_protocol_through_rccl_world_size_one():
    env = dict(os.environ)
    env.pop("BBB_FORCE_COMBINE", None)
    p = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    res = json.loads(line[0][7:])
    for lt, r in res.items():
        # one rank holds every unit, so log(sum over ranks) is the rank's own block: equal up to the extra logsumexp's rounding
        assert r["finite"] and r["fresh"], (lt, r)
        assert r["d_eager"] <= 3e-6 * max(r["scale"], 1.0) and r["kl_eager"] <= 1e-6, (lt, r)
        assert r["d_graph"] <= 3e-6 * max(r["scale"], 1.0) and r["d_kl"] <= 1e-6, (lt, r)
        assert r["d_group"] <= 3e-6 * max(r["scale"], 1.0) and r["d_group_kl"] <= 1e-6 and r["group_same"], (lt, r)
        # the recorded collective (one host call per step) and the eager one are the same arithmetic
        assert r["same_protocols"] and r["fused"]["plain"] == [False, False] and r["fused"]["rccl_eager"] == [False, False], (lt, r)
        assert r["fused"]["rccl"] == [True, True], "RCCL refused to record all_gather_into_tensor into a hipGraph: %r" % (r["fused"],)


_RANK_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "pytorch-bayesiancnn_amd")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import torch, torch.distributed as dist
import ref_port_torch as P
from bbb_hip import ensemble, zoo, rng
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("BBB_TEST_BACKEND", "nccl")                 # "gloo" + BBB_TEST_DEVICE=0: several ranks on ONE device (rehearsal)
dev = torch.device("cuda", int(os.environ.get("BBB_TEST_DEVICE", os.environ["LOCAL_RANK"]))); torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group(backend)
group = dist.group.WORLD
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device=dev)
E = 10
out = {"world": world}
with torch.no_grad():
    rng.manual_seed(7, 100)
    lo1, kl1 = ensemble.mc_forward(net, x, E)                       # this GPU alone: the single-device step
    rng.manual_seed(7, 100)
    loN, klN = ensemble.mc_forward(net, x, E, group=group)          # sharded over the ranks, eager
    rng.manual_seed(7, 100)
    pipe = ensemble.GraphedPipeline(net, x, E, depth=2, group=group)
    g_steps = []
    for _ in range(4):
        lo, kl = pipe.step()
        pipe.sync()
        g_steps.append((lo.clone(), kl.clone()))
    rng.manual_seed(7, 100)
    e_steps = [ensemble.mc_forward(net, x, E) for _ in range(4)]
    # groups of 4 steps per launch dealt to the ranks as contiguous draw ranges (what bench.py runs at N > 1)
    del pipe
    xs = [torch.rand(512, 3, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(11 + i)) for i in range(4)]
    rng.manual_seed(7, 100)
    pipe = ensemble.GraphedPipeline(net, x, E, depth=2, group=group, steps_per_launch=4)
    views = [pipe.step(xs[i]) for i in range(4)]
    pipe.sync()
    grp_steps = [(a.clone(), b.clone()) for a, b in views]
    rng.manual_seed(7, 100)
    grp_ref = [ensemble.mc_forward(net, xs[i], E) for i in range(4)]
S, lo_u, hi_u = ensemble.shard_plan(net, x, E, rank, world)
# every rank must hold the same bits: gather rank 0's result and compare
ref = loN.clone(); dist.broadcast(ref, 0, group=group)
out.update(units=hi_u - lo_u, S=S, fused=[bool(l.fused) for l in pipe.lanes], same_on_all_ranks=bool(torch.equal(ref, loN)),
           d_eager=(loN - lo1).abs().max().item(), kl_rel=abs(klN.item() - kl1.item()) / abs(kl1.item()),
           d_graph=max((a[0] - b[0]).abs().max().item() for a, b in zip(g_steps, e_steps)),
           kl_graph=max(abs(a[1].item() - b[1].item()) / abs(b[1].item()) for a, b in zip(g_steps, e_steps)),
           d_group=max((a[0] - b[0]).abs().max().item() for a, b in zip(grp_steps, grp_ref)),
           kl_group=max(abs(a[1].item() - b[1].item()) / abs(b[1].item()) for a, b in zip(grp_steps, grp_ref)),
           scale=lo1.abs().max().item())
out["d_graph"] = max(out["d_graph"], out["d_group"]); out["kl_graph"] = max(out["kl_graph"], out["kl_group"])
flags = torch.tensor([float(out["same_on_all_ranks"]), out["d_eager"], out["d_graph"]], device=dev)
dist.all_reduce(flags[:1], op=dist.ReduceOp.MIN, group=group); dist.all_reduce(flags[1:], op=dist.ReduceOp.MAX, group=group)
dist.barrier(group=group)
dist.destroy_process_group()
if rank == 0:
    out.update(all_same=bool(flags[0].item() > 0.5), d_eager_max=flags[1].item(), d_graph_max=flags[2].item())
    print("RESULT " + json.dumps(out))
'''


def _run_ranks(n, port, **extra_env):
    import tempfile
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("BBB_FORCE_COMBINE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, "worker.py")
        open(w, "w").write(_RANK_WORKER % {"root": ROOT})
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                            "127.0.0.1", "--master-port", str(port), w], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    r = json.loads(line[0][7:])
    assert r["world"] == n and r["all_same"], r
    # the rank-order log-sum-exp over blocks vs the one-GPU tail: same draws, one more logsumexp's rounding
    assert r["d_eager_max"] <= 3e-6 * max(r["scale"], 1.0) and r["kl_rel"] <= 1e-6, r
    assert r["d_graph_max"] <= 3e-6 * max(r["scale"], 1.0) and r["kl_graph"] <= 1e-6, r
    return r


def test_rank_worker_under_the_launcher_at_world_size_one():
    """The multi-GPU test's worker, launched exactly as below with ONE rank: keeps the worker and its launch line exercised on
    the one-GPU boxes the suite usually runs on."""
    r = _run_ranks(1, 29559)
    assert r["units"] == 10 and r["S"] == 1


@pytest.mark.parametrize("n", [2, 3])
def test_rank_worker_with_several_ranks_on_one_device_over_gloo(n):
    """The multi-GPU worker with 2 and 3 REAL ranks (processes), all on device 0, gloo instead of RCCL (which does not allow two
    ranks per device): the (draw x batch-slice) work units of one step AND the groups of four steps dealt out as whole draws
    (3 ranks: 40 draws as 14 + 13 + 13, shares that start and end in the middle of a step) give every rank the single-device
    results -- eager collective protocol (gloo cannot be recorded into a hipGraph)."""
    r = _run_ranks(n, 29563 + n, BBB_TEST_BACKEND="gloo", BBB_TEST_DEVICE="0")
    assert not any(r["fused"]), r


def test_sharded_step_over_rccl_on_every_gpu_of_the_box():
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("needs >= 2 GPUs (RCCL does not allow two ranks on one device); %d visible" % n)
    r = _run_ranks(n, 29561)
    assert all(r["fused"]), r                              # rank 0's lanes recorded their collective into the step's graph
