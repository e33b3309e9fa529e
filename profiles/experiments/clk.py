"""Shader clock under load, un-profiled: a one-wave probe (s_memtime vs the 100 MHz s_memrealtime) runs on a side stream while the
fp32 GEMM layers run back-to-back on the main stream."""
import ctypes, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'pytorch-bayesiancnn_amd'))
import torch
from bbb_hip import ops
lib = ctypes.CDLL(os.path.join(HERE, 'libclk.so'))
lib.clk_probe.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
B = 512
L = {"conv1": (3, 32, 32, 64, 11, 4, 5), "conv2": (64, 4, 4, 192, 5, 1, 2), "conv3": (192, 2, 2, 384, 3, 1, 1), "conv4": (384, 2, 2, 256, 3, 1, 1),
     "conv5": (256, 2, 2, 128, 3, 1, 1)}
side = torch.cuda.Stream()
out = torch.zeros(2, dtype=torch.int64, device='cuda')
def probe(iters):
    with torch.cuda.stream(side):
        lib.clk_probe(out.data_ptr(), iters, torch.cuda.current_stream().cuda_stream)
def clock():
    side.synchronize()
    t, w = out.tolist()
    return round(t / w * 100.0 / 1e3, 3), round(w / 100.0, 1)      # GHz, probe duration in us
res = {}
probe(20000); torch.cuda.synchronize(); res["idle"] = clock()
zero = bool(os.environ.get("ZERO"))
for E in (10, 40):
    for name, (Cin, H, W, Cout, k, st, pd) in L.items():
        x = torch.randn(1 if name == "conv1" else E, Cin, H, W, B, device='cuda')
        w = torch.randn(E, Cout, Cin, k, k, device='cuda') * 0.05
        b = torch.randn(E, Cout, device='cuda')
        if zero: x.zero_(); w.zero_()
        fn = lambda: ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # ~3 graph replays of load; the probe starts after the first and ends before the last
        s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 40
        g.replay(); probe(int(us * 20 * 1.2 / 0.5)); g.replay(); g.replay(); g.replay(); torch.cuda.synchronize()
        res[f"E{E}_{name}"] = [round(us, 1)] + list(clock())
        del x, w, b, g
print(json.dumps(res))
