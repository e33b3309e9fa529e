"""Split-fp16 GEMM mode on the other fp32 BBB configurations: E = 25 (configs[3]) and 224x224 x 512 images (configs[4]); ms per
step with three lanes, both modes, and the largest log-probability difference between the modes under the same noise."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble, ops, rng
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for name in ("configs[3]", "configs[4]"):
    cfg = bench.CONFIGS[name]
    net, x = bench.build_net(cfg, dev)
    row = {"config": name}
    outs = {}
    for mode in ("fp32", "fp16x2"):
        ops.gemm_mode = mode
        with torch.no_grad():
            rng.manual_seed(3, call=0)
            outs[mode] = ensemble.mc_forward(net, x, cfg["E"])[0].clone()
            pipe = ensemble.GraphedPipeline(net, x, cfg["E"], depth=3)
            n = 60 if cfg["hw"] == 32 else 12
            for _ in range(6):
                pipe.step()
            pipe.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            pipe.sync()
            row[mode + "_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
            del pipe
    row["max_abs_diff"] = float((outs["fp16x2"] - outs["fp32"]).abs().max())
    row["max_abs"] = float(outs["fp32"].abs().max())
    print(json.dumps(row), flush=True)
ops.gemm_mode = "fp32"
