"""Where the persistent chain kernel's workgroups spend their time (needs a -DPCHAIN_PROFILE build of the library:
    BBB_HIP_LIB=scratch/r3/prof/libbbb_hip_prof.so python profiles/experiments/chain_prof.py).
Per workgroup: 100 MHz ticks spent dequeuing, waiting for dependencies, executing items, publishing; kernel span."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch  # noqa: E402
from bbb_hip import ensemble, ops, rng, zoo, _lib  # noqa: E402

PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRI, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device=dev)
key = (dev.index, "chain", _lib.cur_stream(dev))
ops._scratch[key] = torch.zeros(1 << 17, dtype=torch.int32, device=dev)
ensemble.use_chain = True
for E, group in ((10, 0), (10, 1), (10, 2), (10, 0x300), (1, 0), (1, 2), (25, 0)):
    ensemble.chain_flags = group
    with torch.no_grad():
        for _ in range(3):
            ensemble._mc_logits_chwn(net, x, E, 7, 3)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ensemble._mc_logits_chwn(net, x, E, 7, 3)
        e.record()
        torch.cuda.synchronize()
    ws = ops._scratch[key].cpu()
    off = (32 + 12 * 64 + 8 * 32 + 12 * 8 * 64 + 1024 + 15) // 16 * 16
    p = ws[off:off + 1024 * 16].view(torch.int64).view(1024, 8).double()
    tick = 0.01   # us per tick (100 MHz)
    span = (p[:, 7].max() - p[:, 6].min()).item() * tick
    row = {"E": E, "flags": hex(group), "step_us_eager_events": round(s.elapsed_time(e) * 1e3, 1), "kernel_span_us": round(span, 1)}
    for i, name in enumerate(("dequeue", "dep_wait", "exec", "publish")):
        row[name + "_us_mean"] = round(p[:, i].mean().item() * tick, 1)
        row[name + "_us_max"] = round(p[:, i].max().item() * tick, 1)
    row["items_mean"] = round(p[:, 4].mean().item(), 2)
    row["waits_mean"] = round(p[:, 5].mean().item(), 2)
    row["wg_lifetime_us_mean"] = round((p[:, 7] - p[:, 6]).mean().item() * tick, 1)
    row["start_skew_us"] = round((p[:, 6].max() - p[:, 6].min()).item() * tick, 1)
    row["per_xcd_exec_us"] = [round(p[x_::8, 2].mean().item() * tick, 1) for x_ in range(8)]
    print(json.dumps(row), flush=True)
