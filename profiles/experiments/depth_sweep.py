"""Steps in flight (GraphedPipeline depth) vs throughput for every BASELINE configuration: ms per step at depth 1..8."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from bbb_hip import ensemble  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for name in sys.argv[1:] or list(bench.CONFIGS):
    cfg = bench.CONFIGS[name]
    net, x = bench.build_net(cfg, dev)
    row = {"config": name}
    for depth in (1, 2, 3, 4, 5, 6, 8, 12):
        with torch.no_grad():
            pipe = ensemble.GraphedPipeline(net, x, cfg["E"], depth=depth, precision=cfg["precision"]) if depth > 1 else \
                ensemble.GraphedMC(net, x, cfg["E"], precision=cfg["precision"])
            n = 400 if cfg["hw"] == 32 else 40
            for _ in range(n // 4):
                pipe.step()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(n):
                    pipe.step()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
            del pipe
        row[f"d{depth}_ms"] = round(best * 1e3, 4)
    print(json.dumps(row), flush=True)
