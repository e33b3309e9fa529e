"""Per-layer timing of the batch-innermost fp32 GEMM on the AlexNet/CIFAR shapes (bs 512), for kernel-variant experiments:
BBB_HIP_LIB=<variant .so> python scratch/r2/layers.py <tag> [E ...]  -> one JSON line (us and TFLOP/s of in-bounds work per layer,
an int32 checksum of every output so that variants can be compared bitwise)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops
B = 512
L = [("conv1", 3, 32, 32, 64, 11, 4, 5, 1.256e9), ("conv2", 64, 4, 4, 192, 5, 1, 2, 2.465e9), ("conv3", 192, 2, 2, 384, 3, 1, 1, 1.208e9),
     ("conv4", 384, 2, 2, 256, 3, 1, 1, 1.611e9), ("conv5", 256, 2, 2, 128, 3, 1, 1, 0.537e9), ("fc", 128, 1, 1, 10, 1, 1, 0, 1.31e6)]
def t(fn, n=20, reps=5):
    if os.environ.get('EAGER'):
        for _ in range(6): fn()
        torch.cuda.synchronize(); return 1.0
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best
out = {"tag": sys.argv[1] if len(sys.argv) > 1 else "", "E": {}}
for E in [int(a) for a in sys.argv[2:]] or [10, 40, 2]:
    torch.manual_seed(0)
    row = {}; tot = 0.0; fl = 0.0
    for name, Cin, H, W, Cout, k, st, pd, flops in L:
        x = torch.randn(1 if name == "conv1" else E, Cin, H, W, B, device='cuda')
        w = torch.randn(E, Cout, Cin, k, k, device='cuda') * 0.05
        b = torch.randn(E, Cout, device='cuda')
        if os.environ.get('ZERO'): x.zero_(); w.zero_()
        y = ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
        us = t(lambda: ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus"))
        row[name] = [round(us, 1), round(flops * E / us / 1e6, 1), int(y.view(torch.int32).sum(dtype=torch.int64).item())]
        tot += us; fl += flops * E
        del x, w, b, y
    row["total_us"] = round(tot, 1); row["TF"] = round(fl / tot / 1e6, 1)
    out["E"][E] = row
print(json.dumps(out))
