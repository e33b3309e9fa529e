import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops
torch.manual_seed(0)
B = 512
L = [("conv2", 64, 4, 4, 192, 5, 1, 2), ("conv3", 192, 2, 2, 384, 3, 1, 1), ("conv4", 384, 2, 2, 256, 3, 1, 1), ("conv5", 256, 2, 2, 128, 3, 1, 1), ("fc", 128, 1, 1, 10, 1, 1, 0)]
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
orig = ops._split_plan
for E in (1, 2):
    for name, Cin, H, W, Cout, k, st, pd in L:
        x = torch.randn(E, Cin, H, W, B, device='cuda'); w = torch.randn(E, Cout, Cin, k, k, device='cuda') * 0.05; b = torch.zeros(E, Cout, device='cuda')
        row = {}
        for S in (1, 2, 3, 4, 6, 8):
            ops._split_plans.clear()
            def forced(d, lrt, S=S):
                ks, wsb, nt = orig(d, lrt)
                if S == 1: return (1, 0, 0)
                G = ((d.cout + 63) // 64) * d.draws
                ho = (d.h + 2 * d.pad_h - d.dil_h * (d.kh - 1) - 1) // d.stride_h + 1
                wo = (d.w + 2 * d.pad_w - d.dil_w * (d.kw - 1) - 1) // d.stride_w + 1
                items = G * ho * wo * ((d.batch + 63) // 64)
                return (S, items * S * 16384, items)
            ops._split_plan = forced
            try:
                row[S] = round(t(lambda: ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")), 1)
            except Exception as ex:
                row[S] = str(ex)[:40]
        ops._split_plan = orig
        print(E, name, row, flush=True)
