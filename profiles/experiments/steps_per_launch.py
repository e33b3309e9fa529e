"""One-draw steps: G steps per launch (GraphedPipeline steps_per_launch) x lanes, ms per step, for BASELINE configs[1] (bf16
3Conv3FC bs 256), configs[2] (LRT AlexNet CIFAR-100 bs 512) and fp32 BBB AlexNet bs 512 at one draw."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def ms_per_step(net, x, precision, G, depth, n=480):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, 1, depth=depth, precision=precision, steps_per_launch=G)
        for _ in range(n // 4):
            pipe.step()
        pipe.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            pipe.sync()
            best = min(best, (time.perf_counter() - t0) / n)
    del pipe
    return round(best * 1e3, 4)


cases = [("configs[1]", dict(bench.CONFIGS["configs[1]"])), ("configs[2]", dict(bench.CONFIGS["configs[2]"])),
         ("alexnet bbb fp32 bs512 E=1", dict(bench.CONFIGS["metric"], E=1))]
for name, cfg in cases:
    net, x = bench.build_net(cfg, dev)
    for G in (1, 2, 4, 8):
        row = {"config": name, "G": G}
        for depth in (1, 2, 3, 4):
            row[f"lanes{depth}"] = ms_per_step(net, x, cfg["precision"], G, depth)
        print(json.dumps(row), flush=True)
