import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops, ensemble
torch.manual_seed(0)
def t(fn, n=20, reps=5):
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
CIFAR = [("conv1", 3, 32, 32, 64, 11, 4, 5), ("conv2", 64, 4, 4, 192, 5, 1, 2), ("conv3", 192, 2, 2, 384, 3, 1, 1), ("conv4", 384, 2, 2, 256, 3, 1, 1), ("conv5", 256, 2, 2, 128, 3, 1, 1)]
BIG = [("conv1", 3, 224, 224, 64, 11, 4, 5), ("conv2", 64, 28, 28, 192, 5, 1, 2), ("conv3", 192, 14, 14, 384, 3, 1, 1), ("conv4", 384, 14, 14, 256, 3, 1, 1), ("conv5", 256, 14, 14, 128, 3, 1, 1)]
C3 = [("conv1", 3, 32, 32, 32, 5, 1, 2), ("conv2", 32, 15, 15, 64, 5, 1, 2), ("conv3", 64, 7, 7, 128, 5, 1, 1), ("fc1", 512, 1, 1, 1000, 1, 1, 0), ("fc2", 1000, 1, 1, 1000, 1, 1, 0), ("fc3", 1000, 1, 1, 10, 1, 1, 0)]
for tag, L, E, B in (("3c3fc", C3, 1, 256), ("cifar", CIFAR, 1, 512), ("cifar", CIFAR, 2, 512), ("cifar", CIFAR, 10, 512), ("cifar", CIFAR, 40, 512), ("224", BIG, 1, 64)):
    tot = 0.0; fl_tot = 0.0; row = {}
    for name, Cin, H, W, Cout, k, st, pd in L:
        K = Cin * k * k; Kp = (K + 7) & ~7
        TM = Cin % 8 == 0
        x = torch.randn(1 if name == "conv1" else E, Cin, H, W, B, device='cuda').to(torch.bfloat16)
        w = (torch.randn(E, Cout, Kp, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(E, Cout, device='cuda')
        us = t(lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (Cin, k, k), st, pd, 1, act="softplus", tap_major=TM))
        fl = ensemble.conv_flops(B, Cin, H, W, Cout, k, k, st, pd, 1, E)[0]
        yv = ops.conv2d_chwn_bf16_forward(x, w, b, (Cin, k, k), st, pd, 1, act="softplus", tap_major=TM)
        row[name] = [round(us, 1), round(fl / us / 1e6, 1), int(yv.view(torch.int16).sum(dtype=torch.int64).item()) % 100000]
        tot += us; fl_tot += fl
        del x, w, b
    row["total_us"] = round(tot, 1); row["TF"] = round(fl_tot / tot / 1e6, 1)
    print(tag, "E", E, "B", B, json.dumps(row), flush=True)
