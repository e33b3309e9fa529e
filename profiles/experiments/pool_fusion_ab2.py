"""Pool fusion rule, second A/B: off / rule / forced on the launch shapes the rule excludes -- one step per launch (one and three
lanes), num_ens 25, the 224x224 shard -- two interleaved rounds."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ensemble, ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)


def ms_per_step(net, x, E, G, depth, n):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=depth, steps_per_launch=G) if (depth > 1 or G > 1) else ensemble.GraphedMC(net, x, E)
        sync = pipe.sync if hasattr(pipe, "sync") else (lambda: torch.cuda.synchronize())
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(G * depth): pipe.step()
            sync()
        vals = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n): pipe.step()
            sync(); vals.append((time.perf_counter() - t0) / n)
    del pipe
    return round(statistics.median(vals) * 1e3, 4)


nets = {k: bench.build_net(bench.CONFIGS[k], dev) for k in ("metric", "configs[3]", "configs[4]")}
modes = {"off": (False, 2048, 1.03), "rule": (True, 2048, 1.03), "forced": (True, 0, 2.0)}
for rnd in range(2):
    for tag, (on, mn, imb) in modes.items():
        ops.pool_fusion, ops.pool_fuse_min_items, ops.pool_fuse_imbalance = on, mn, imb
        net, x = nets["metric"]
        row = {"pool_fusion": tag, "metric_G1x1": ms_per_step(net, x, 10, 1, 1, 120), "metric_G1x3": ms_per_step(net, x, 10, 1, 3, 240),
               "metric_G2x2": ms_per_step(net, x, 10, 2, 2, 240)}
        net, x = nets["configs[3]"]
        row["E25_G1x3"] = ms_per_step(net, x, 25, 1, 3, 90)
        net, x = nets["configs[4]"]
        row["a224_G1x3"] = ms_per_step(net, x, bench.CONFIGS["configs[4]"]["E"], 1, 3, 30)
        print(json.dumps(row), flush=True)
