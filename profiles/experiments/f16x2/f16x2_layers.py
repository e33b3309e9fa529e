"""Split-fp16 contraction vs the fp32 MFMA kernel: us per conv / linear launch of the metric step (AlexNet bs 512, E = 10; 20
launches per hipGraph), then ms per step with 1 and 3 lanes, both modes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble, ops
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = dict(bench.CONFIGS["metric"])
net, x = bench.build_net(cfg, dev)


def per_launch(E):
    rec = bench.LaunchRecorder()
    with torch.no_grad():
        ensemble._mc_logits_chwn(net, x, E, 7, 3, timers=rec)
    torch.cuda.synchronize()
    agg = rec.time_in_graphs(dev)
    return [round(u, 1) for u in rec.per_launch_us], round(agg["conv_gemm"]["work"] / (agg["conv_gemm"]["ms"] * 1e-3) / 1e12, 1)


def ms_per_step(E, lanes, n=200):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=lanes)
        for _ in range(30):
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


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
if quick:
    ops.gemm_mode = "fp16x2"
    us, tf = per_launch(10)
    print(json.dumps({"E": 10, "mode": "fp16x2", "us_per_launch": us}), flush=True)
    sys.exit(0)
for E in (10, 1):
    for mode in ("fp32", "fp16x2"):
        ops.gemm_mode = mode
        us, tf = per_launch(E)
        print(json.dumps({"E": E, "mode": mode, "us_per_launch": us, "fp32_equivalent_TFLOPs": tf,
                          "ms_1lane": ms_per_step(E, 1), "ms_3lanes": ms_per_step(E, 3)}), flush=True)
ops.gemm_mode = "fp32"
