"""Round 4: the range-free split-bf16 contraction (bbb_conv2d_chwn_bf16x3_fwd) vs the fp32 MFMA kernel and round 3's split-fp16
form: accuracy against float64 on a few operand scales, us per conv / linear launch of the metric step (AlexNet bs 512, E = 10; 20
launches per hipGraph), ms per step with 1 and 3 lanes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import bench
import bbb_numpy as O
from bbb_hip import ensemble, ops
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
MODES = ["fp32", "bf16x3"]        # (round 3's split-fp16 form, measured in the same script before it was removed: profiles/r04_bf16x3.txt)


def accuracy():
    B, Cin, H, W, Cout, k, pd = 256, 64, 4, 4, 192, 5, 2
    for xs, ws in ((3.0, 0.2), (0.25, 0.01), (100.0, 8.0), (0.004, 0.0003), (1e-6, 1e-9), (1e6, 1e-12)):
        torch.manual_seed(1)
        x = torch.randn(1, Cin, H, W, B, device=dev) * xs
        w = torch.randn(1, Cout, Cin, k, k, device=dev) * ws
        b = torch.randn(1, Cout, device=dev) * xs * ws
        xe = x[0].permute(3, 0, 1, 2).double().cpu().numpy()
        want = O.conv2d(xe, w[0].double().cpu().numpy(), b[0].double().cpu().numpy(), 1, pd, 1)
        mag = O.conv2d(np.abs(xe), np.abs(w[0].double().cpu().numpy()), np.abs(b[0].double().cpu().numpy()), 1, pd, 1)
        row = {"x_scale": xs, "w_scale": ws}
        for mode in MODES:
            ops.gemm_mode = mode
            saved = ops.bf16x3_min_workgroups
            ops.bf16x3_min_workgroups = 0
            y = ops.conv2d_chwn_forward(x, w, b, 1, pd, 1)
            ops.bf16x3_min_workgroups = saved
            row[mode] = float("%.3g" % float((np.abs(y[0].permute(3, 0, 1, 2).double().cpu().numpy() - want) / mag).max()))
        print(json.dumps(row), flush=True)
    ops.gemm_mode = "fp32"


def per_launch(net, x, E, mode):
    rec = bench.LaunchRecorder()
    with torch.no_grad():
        ensemble._mc_logits_chwn(net, x, E, 7, 3, timers=rec, precision=mode)
    torch.cuda.synchronize()
    agg = rec.time_in_graphs(dev)
    return [round(u, 1) for u in rec.per_launch_us], round(agg["conv_gemm"]["work"] / (agg["conv_gemm"]["ms"] * 1e-3) / 1e12, 1)


def ms_per_step(net, x, E, lanes, mode, n=200):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=lanes, precision=mode)
        bench.preheat(pipe.step, 0.3, dev)
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


accuracy()
cfg = dict(bench.CONFIGS["metric"])
net, x = bench.build_net(cfg, dev)
for E in (10, 1):
    for mode in MODES:
        ops.gemm_mode = "fp32"
        us, tf = per_launch(net, x, E, mode)
        print(json.dumps({"E": E, "mode": mode, "us_per_launch": us, "fp32_equivalent_TFLOPs": tf,
                          "ms_1lane": ms_per_step(net, x, E, 1, mode), "ms_3lanes": ms_per_step(net, x, E, 3, mode)}), flush=True)
