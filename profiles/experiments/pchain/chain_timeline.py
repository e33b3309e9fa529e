"""Per-item timeline of the persistent chain kernel (PCHAIN_PROFILE build): when each stage starts / ends, how many items run
at a time, where the machine idles.  BBB_HIP_LIB=scratch/r3/prof/libbbb_hip_prof.so python profiles/experiments/chain_timeline.py"""
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
ops._scratch[key] = torch.zeros(1 << 19, dtype=torch.int32, device=dev)
ensemble.use_chain = True
names = ["conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "conv5", "pool3", "fc"]
for E, flags in ((10, 1), (10, 0), (1, 0)):
    ensemble.chain_flags = flags
    with torch.no_grad():
        for _ in range(3):
            ensemble._mc_logits_chwn(net, x, E, 7, 3)
        torch.cuda.synchronize()
    ws = ops._scratch[key].cpu()
    off = (32 + 12 * 64 + 8 * 32 + 12 * 8 * 64 + 1024 + 15) // 16 * 16
    p = ws[off:off + 1024 * 16].view(torch.int64).view(1024, 8)
    lg = ws[off + 2048 * 16: off + 2048 * 16 + 1024 * 32 * 4].view(1024, 32, 4).long() & 0xFFFFFFFF
    n_items = p[:, 4].clamp(max=30)
    t0 = int(p[:, 6].min()) & 0xFFFFFFFF
    rows = []
    for b in range(1024):
        for i in range(int(n_items[b])):
            s_, e_, st, sl = [int(v) for v in lg[b, i]]
            rows.append(((s_ - t0) & 0xFFFFFFFF, (e_ - t0) & 0xFFFFFFFF, st, sl, b))
    tick = 0.01
    end = max(r[1] for r in rows) * tick
    print(json.dumps({"E": E, "flags": flags, "items_logged": len(rows), "span_us": round(end, 1)}))
    for st in range(9):
        rs = [r for r in rows if r[2] == st]
        if not rs:
            continue
        durs = sorted((r[1] - r[0]) * tick for r in rs)
        print(json.dumps({"stage": names[st], "items": len(rs), "first_start_us": round(min(r[0] for r in rs) * tick, 1),
                          "last_start_us": round(max(r[0] for r in rs) * tick, 1), "last_end_us": round(max(r[1] for r in rs) * tick, 1),
                          "dur_us_p10_med_p90_max": [round(durs[len(durs) // 10], 1), round(durs[len(durs) // 2], 1),
                                                     round(durs[len(durs) * 9 // 10], 1), round(durs[-1], 1)],
                          "sum_dur_ms": round(sum(durs) / 1e3, 2)}))
    nb = int(end // 20) + 1
    running = [0.0] * nb
    for s_, e_, st, sl, b in rows:
        a, c = s_ * tick, e_ * tick
        k = int(a // 20)
        while k * 20 < c and k < nb:
            lo, hi = max(a, k * 20), min(c, (k + 1) * 20)
            running[k] += (hi - lo) / 20
            k += 1
    print(json.dumps({"running_items_per_20us": [round(v) for v in running]}))
