"""Round 4: pooling in the GEMM launch (serial window, ops.pool_fusion) A/B on one box: ms per step of the metric pipeline
(G = 4 x 2 lanes) with the fusion rule on / off, three interleaved rounds, plus the per-launch GEMM times and, with the rule forced
(min items 0, imbalance 2.0), conv2's pooled form as well."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ensemble, ops, rng
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]; net, x = bench.build_net(cfg, dev); E = cfg["E"]


def ms_per_step(G, depth, n=240):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=depth, steps_per_launch=G)
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            for _ in range(G * depth): pipe.step()
            pipe.sync()
        vals = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n): pipe.step()
            pipe.sync(); vals.append((time.perf_counter() - t0) / n)
    del pipe
    return round(statistics.median(vals) * 1e3, 4)


def launches(G):
    rec = bench.LaunchRecorder(); rec.reps = 10
    xg = x.repeat(G, 1, 1, 1) if G > 1 else x
    with torch.no_grad():
        seed, call0 = rng.next_calls(G * E)
        ensemble._local_lse(net, xg, E, seed, call0, E, timers=rec, groups=G) if G > 1 else ensemble.mc_forward(net, x, E, timers=rec)
        torch.cuda.synchronize(); rec.time_in_graphs(dev)
    return rec.per_launch_us


modes = {"off": (False, 2048, 1.03), "rule (conv1)": (True, 2048, 1.03), "forced (conv1 + conv2)": (True, 0, 2.0)}
for rnd in range(3):
    for tag, (on, mn, imb) in modes.items():
        ops.pool_fusion, ops.pool_fuse_min_items, ops.pool_fuse_imbalance = on, mn, imb
        row = {"pool_fusion": tag, "G4x2_ms_per_step": ms_per_step(4, 2)}
        if rnd == 0:
            row["gemm_per_launch_us_G4"] = launches(4)
            row["G1x3_ms_per_step"] = ms_per_step(1, 3)
        print(json.dumps(row), flush=True)
