"""Effective bandwidth of the layout passes of the training step's weight gradients at the metric shape (bs 512 x 10 draws, BBB,
BayesianAlexNet): chwn_to_bhwc(x), chwn_grad_as_weights(g), and the tap transpose of the finished gradient -- per layer, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops

E, B = int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 512
LAYERS = [("conv2", 64, 192, 4, 25), ("conv3", 192, 384, 2, 9), ("conv4", 384, 256, 2, 9), ("conv5", 256, 128, 2, 9), ("fc", 128, 10, 1, 1)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


tot = 0.0
for name, cin, cout, hw, taps in LAYERS:
    x = torch.rand(E, cin, hw, hw, B, device="cuda")
    g = torch.rand(E, cout, hw, hw, B, device="cuda")
    y = torch.rand(E, cout, taps, cin, device="cuda")
    gw = torch.empty(E, cout, cin, taps, device="cuda")
    t1 = timed(lambda: ops.chwn_to_bhwc(x))
    t2 = timed(lambda: ops.chwn_grad_as_weights(g))
    t3 = timed(lambda: ops._transpose_sum_batched(y, gw, taps, cin, (E, cout, 1), (cout * taps * cin, taps * cin, 0), (cout * cin * taps, cin * taps, 0), cin, taps, 1, 0))
    tot += t1 + t2 + t3
    print("%-6s x->bhwc %7.1f us %5.2f TB/s | g->weights %7.1f us %5.2f TB/s | taps %7.1f us %5.2f TB/s" % (
        name, t1, 8 * x.numel() / t1 / 1e6, t2, 8 * g.numel() / t2 / 1e6, t3, 8 * y.numel() / t3 / 1e6))
print("total %.1f us per step" % tot)
