"""The N > 1 step protocol through RCCL itself (backend "nccl") on ONE MI355X: a world-size-1 process group with the sharded
code path forced (BBB_FORCE_COMBINE=1) -- graph replay, ONE all_gather_into_tensor on the lane's stream, the reduction over
ranks as a second small graph -- must reproduce the plain single-process step.  (More than one rank per device is not something
RCCL allows; world sizes 2-8 are covered with gloo in test_host_cpu.py and with simulated ranks in test_gpu_sharding.py.)
Runs in a subprocess: a process group is process-global state.  Run with -m gpu."""
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
    # graph lanes: plain vs the forced multi-rank protocol (send buffer, all_gather, post graph)
    res = {}
    for tag, grp, force in (("plain", None, "0"), ("rccl", group, "1")):
        os.environ["BBB_FORCE_COMBINE"] = force
        rng.manual_seed(7, 200)
        pipe = ensemble.GraphedPipeline(net, x, E, depth=2, group=grp)
        steps = []
        for _ in range(5):
            lo, kl = pipe.step()
            pipe.sync()
            steps.append((lo.clone(), kl.clone()))
        res[tag] = steps
        del pipe
    os.environ["BBB_FORCE_COMBINE"] = "0"
    d_eager = (lo1 - lo0).abs().max().item()
    d_graph = max((a[0] - b[0]).abs().max().item() for a, b in zip(res["plain"], res["rccl"]))
    d_kl = max(abs(a[1].item() - b[1].item()) / abs(a[1].item()) for a, b in zip(res["plain"], res["rccl"]))
    fresh = not torch.equal(res["rccl"][0][0], res["rccl"][1][0])
    out[lt] = dict(d_eager=d_eager, kl_eager=abs(kl1.item() - kl0.item()) / abs(kl0.item()), d_graph=d_graph, d_kl=d_kl, fresh=fresh,
                   scale=lo0.abs().max().item(), finite=bool(torch.isfinite(res["rccl"][-1][0]).all()))
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
