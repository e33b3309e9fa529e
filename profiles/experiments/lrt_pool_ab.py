"""Round 5 (review item 5): the pooled LRT launch (pconv_body.cuh POOL + LRT) against LRT conv + maxpool_chwn(2, 2), AlexNet conv1 /
conv2 at bs 512, G one-draw steps per launch (configs[2]'s launches).  us per launch, hot, in hipGraphs of 10."""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)


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


B = 512
with torch.no_grad():
    for G in (4, 16):
        for name, (Cin, H, Cout, k, s, p) in {"conv1": (3, 32, 64, 11, 4, 5), "conv2": (64, 4, 192, 5, 1, 2)}.items():
            x = torch.rand(G, Cin, H, H, B, device=dev)
            w_mu = torch.randn(Cout, Cin, k, k, device=dev) * 0.1
            w_var = torch.rand(Cout, Cin, k, k, device=dev) * 1e-4
            b_mu, b_var = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) * 1e-4
            un = hot_us(lambda: ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 1, 0, 3, s, p, 1, act="softplus")[0])
            y = ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 1, 0, 3, s, p, 1, act="softplus")[0]
            pl = hot_us(lambda: ops.maxpool_chwn(y, 2, 2))
            fu = hot_us(lambda: ops.lrt_conv2d_chwn_forward(x, w_mu, w_var, b_mu, b_var, 1, 0, 3, s, p, 1, act="softplus", pool=True)[0])
            print(json.dumps({"G": G, "layer": name, "lrt_conv_us": un, "maxpool_us": pl, "sum_us": round(un + pl, 2), "fused_us": fu}), flush=True)
