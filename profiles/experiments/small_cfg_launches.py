"""Per-launch times (hot, in hipGraphs of 10) of every launch of a G-step group of configs[1] / configs[2]: where a step of the
single-draw configurations goes once launch overhead is amortised.  usage: small_cfg_launches.py configs[1] 16"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ensemble, rng, ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
name, G = sys.argv[1], int(sys.argv[2])
c = bench.CONFIGS[name]
net, x = bench.build_net(c, dev)
E, prec = c["E"], c["precision"]
xg = x.repeat(G, 1, 1, 1)


import statistics


class AllRec:
    """Keeps EVERY bracketed launch of the pass (bench.LaunchRecorder times the GEMMs only)."""
    def __init__(self):
        self.calls = []

    def bracket(self, tag, info, fn):
        out = fn()
        self.calls.append((tag, fn))
        return out


def hot_us(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(reps):
            fn()
    bench.preheat(g.replay, 0.03, dev)
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (3 * reps))
    return round(statistics.median(ts) * 1e3, 2)


with torch.no_grad():
    rec = AllRec()
    seed, call0 = rng.next_calls(G * E)
    ensemble._local_lse(net, xg, E, seed, call0, E, timers=rec, precision=prec, groups=G)
    torch.cuda.synchronize()
    rows = []
    for tag, fn in rec.calls:
        try:
            rows.append((tag, hot_us(fn)))
        except Exception as exc:
            rows.append((tag, "error %s" % type(exc).__name__))
    tot = sum(v for _, v in rows if isinstance(v, float))
    print(json.dumps({"config": name, "G": G, "launches": rows, "sum_us": round(tot, 1), "us_per_step": round(tot / G, 2)}))
