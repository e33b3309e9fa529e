"""Per-phase s_memtime stamps of the fp32 GEMM's k loop, one sampled workgroup per XCD from the middle of the launch
(needs a -DPCONV_STAMPS build of csrc/pconv_gemm.hip, passed via BBB_HIP_LIB)."""
import ctypes, os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops, _lib
lib = ctypes.CDLL(os.environ["BBB_HIP_LIB"])
B = 512
L = [("conv1", 3, 32, 32, 64, 11, 4, 5), ("conv2", 64, 4, 4, 192, 5, 1, 2), ("conv4", 384, 2, 2, 256, 3, 1, 1), ("conv5", 256, 2, 2, 128, 3, 1, 1)]
ops.split_k = False
for E in (1, 10, 40):
    for name, Cin, H, W, Cout, k, st, pd in L:
        x = torch.randn(1 if name == "conv1" else E, Cin, H, W, B, device='cuda'); w = torch.randn(E, Cout, Cin, k, k, device='cuda') * 0.05; b = torch.zeros(E, Cout, device='cuda')
        for _ in range(3): ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 256)()
        lib.bbb_ts_read(buf)
        rows = [[buf[(xc * 4 + wv) * 8 + i] for i in range(8)] for xc in range(8) for wv in range(4)]
        rows = [r for r in rows if r[7]]
        if not rows: continue
        nt = sum(r[6] for r in rows) / len(rows)
        avg = [sum(r[i] / r[6] for r in rows) / len(rows) for i in range(5)]
        tot = sum(r[5] / r[6] for r in rows) / len(rows)
        print(json.dumps({"E": E, "layer": name, "tiles": round(nt, 1), "cycles_per_tile": {"addr(+loads if not ILV)": round(avg[0]), "mma(+ILV loads)": round(avg[1]),
              "barrier1": round(avg[2]), "vmcnt+LDS store": round(avg[3]), "barrier2": round(avg[4]), "loop_total": round(tot)}, "samples": len(rows)}), flush=True)
