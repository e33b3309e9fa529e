"""Workload for the counter passes of the split-bf16 kernel over MFMA-ready operands: AlexNet CIFAR conv2 .. conv5 at bs 512, 40 slabs per
launch, N eager launches each (rocprofv3 --kernel-trace --pmc ... -- python c8x3_pmc.py).  Also prints the box's 16-bit MFMA-loop ceiling."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops

dev = torch.device("cuda:0"); torch.cuda.set_device(0)
LAYERS = {"conv2": ((64, 4, 4), (192, 64, 5, 5), 2), "conv3": ((192, 2, 2), (384, 192, 3, 3), 1),
          "conv4": ((384, 2, 2), (256, 384, 3, 3), 1), "conv5": ((256, 2, 2), (128, 256, 3, 3), 1)}
B, E, N = 512, int(os.environ.get("E", 40)), int(os.environ.get("N", 4))
with torch.no_grad():
    for name, ((C, H, W), (Co, Ci, kh, kw), pad) in LAYERS.items():
        torch.manual_seed(0)
        x = torch.rand(E, C, H, W, B, device=dev)
        w = torch.randn(E, Co, Ci, kh, kw, device=dev) * (1.0 / (Ci * kh * kw) ** 0.5)
        b = torch.randn(E, Co, device=dev) * 0.1
        xc, wt = ops.c8s3_from_f32(x), ops.w_tap_major(w)
        for _ in range(N):
            ops.conv2d_c8x3_forward(xc, wt, b, (kh, kw), 1, pad, 1, act="softplus")
        torch.cuda.synchronize()
try:
    lib = ctypes.CDLL(os.path.join(ROOT, "profiles", "probe", "libmfma_probe.so"))
    v = (ctypes.c_double * 2)()
    lib.probe_mfma_f16_ceiling.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p]
    rc = lib.probe_mfma_f16_ceiling(v, 5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    print("mfma 32x32x16 16-bit loop ceiling: rc", rc, "TFLOP/s %.1f" % v[0], "clock GHz %.3f" % v[1])
except Exception as exc:
    print("probe failed", exc)
