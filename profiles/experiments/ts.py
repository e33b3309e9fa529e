import ctypes, os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops, _lib
lib = ctypes.CDLL(os.environ["BBB_HIP_LIB"])
B = 512
L = [("conv4", 384, 2, 2, 256, 3, 1, 1), ("conv5", 256, 2, 2, 128, 3, 1, 1), ("conv2", 64, 4, 4, 192, 5, 1, 2)]
for E in (1, 10):
    for name, Cin, H, W, Cout, k, st, pd in L:
        x = torch.randn(E, Cin, H, W, B, device='cuda'); w = torch.randn(E, Cout, Cin, k, k, device='cuda') * 0.05; b = torch.zeros(E, Cout, device='cuda')
        for _ in range(3): ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 512)()
        lib.bbb_ts_read(buf)
        rows = []
        for blk in range(8):
            for wv in range(4):
                o = [buf[blk * 64 + wv * 8 + i] for i in range(8)]
                if o[6]: rows.append(o)
        if not rows: continue
        nt = rows[0][6]
        avg = [sum(r[i] for r in rows) / len(rows) / nt for i in range(5)]
        tot = sum(r[5] for r in rows) / len(rows) / nt
        print(f"E={E} {name}: tiles {nt}, cycles per tile: addr/load-issue {avg[0]:.0f}  mma(+ILV loads) {avg[1]:.0f}  barrier1 {avg[2]:.0f}  vmcnt+LDS store {avg[3]:.0f}  barrier2 {avg[4]:.0f}  | loop total {tot:.0f}  (samples {len(rows)})", flush=True)
