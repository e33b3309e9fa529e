"""Round 4: cost of making the split contraction a property of the LAYER.  AlexNet conv4 / conv5 (the two layers the plan splits)
per launch at E = 1 / 5 / 10 / 25 / 40 draws of 512 images, ops.split_k off (plain chain) vs on (cross-workgroup form for small
launches, in-workgroup form otherwise), graph of 20 launches; then whole steps (metric config 1 / 3 lanes, E = 1, E = 25)."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble, ops
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def timed(fn, reps=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(reps):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / reps)
    return round(statistics.median(ts), 2)


LAYERS = {"conv4": (384, 2, 2, 256, 3, 1), "conv5": (256, 2, 2, 128, 3, 1)}
for name, (Cin, H, W, Cout, k, pd) in LAYERS.items():
    for E in (1, 5, 10, 25, 40):
        x = torch.rand(E, Cin, H, W, 512, device=dev)
        w = torch.randn(E, Cout, Cin, k, k, device=dev) * 0.05
        b = torch.randn(E, Cout, device=dev)
        row = {"layer": name, "E": E}
        for mode in (False, True):
            ops.split_k = mode
            row["split_on" if mode else "plain"] = timed(lambda: ops.conv2d_chwn_forward(x, w, b, 1, pd, 1, act="softplus"))
        print(json.dumps(row), flush=True)


def step_ms(cfg, depth, G=1, n=200):
    net, x = bench.build_net(cfg, dev)
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, cfg["E"], depth=depth, precision=cfg["precision"], steps_per_launch=G)
        n = -(-n // (G * depth)) * G * depth
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(G * depth):
                pipe.step()
            pipe.sync()
        vals = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            pipe.sync()
            vals.append((time.perf_counter() - t0) / n)
    del pipe
    return round(statistics.median(vals) * 1e3, 4)


cases = [("metric", bench.CONFIGS["metric"], 1, 1), ("metric", bench.CONFIGS["metric"], 3, 1), ("metric", bench.CONFIGS["metric"], 2, 4),
         ("E=1", dict(bench.CONFIGS["metric"], E=1), 1, 1), ("E=1", dict(bench.CONFIGS["metric"], E=1), 4, 1), ("E=1", dict(bench.CONFIGS["metric"], E=1), 4, 4),
         ("configs[2]", bench.CONFIGS["configs[2]"], 4, 1), ("configs[2]", bench.CONFIGS["configs[2]"], 4, 4),
         ("configs[3]", bench.CONFIGS["configs[3]"], 3, 1)]
for name, cfg, depth, G in cases:
    row = {"config": name, "lanes": depth, "G": G}
    for mode in (False, True):
        ops.split_k = mode
        row["split_on" if mode else "plain"] = step_ms(cfg, depth, G)
    print(json.dumps(row), flush=True)
