"""Round 4: split-bf16 mode, ms per 512 x 10 step for G steps per launch x lanes (pre-heated, median of 5 blocks)."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]
net, x = bench.build_net(cfg, dev)
for mode in ("bf16x3", "fp32"):
    for G, depth in ((1, 1), (1, 3), (2, 2), (4, 1), (4, 2)):
        with torch.no_grad():
            pipe = ensemble.GraphedPipeline(net, x, 10, depth=depth, steps_per_launch=G, precision=mode)
            n = -(-160 // (G * depth)) * G * depth
            bench.preheat(pipe.step, 0.3, dev)
            pipe.sync()
            vals = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(n):
                    pipe.step()
                pipe.sync()
                vals.append((time.perf_counter() - t0) / n)
        del pipe
        print(json.dumps({"mode": mode, "G": G, "lanes": depth, "ms_per_step": round(statistics.median(vals) * 1e3, 4)}), flush=True)
