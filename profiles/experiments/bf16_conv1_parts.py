"""configs[1] first layer (3Conv3FC conv1, bf16, bs 256) at 16 steps per launch: the launch with and without its activation, and the
3x3 / 2 pooling launches that follow conv1 / conv2 / conv3 -- what a pooled (and pool-before-activation) form could save."""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def hot_us(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        for _ in range(reps):
            fn()
    bench.preheat(g.replay, 0.03, dev)
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (3 * reps))
    return round(statistics.median(ts) * 1e3, 2)


out = {"G": G}
with torch.no_grad():
    B = 256
    x = torch.rand(G, 3, 32, 32, B, device=dev).to(torch.bfloat16)
    w = (torch.randn(G, 32, ops.bf16_row_pitch(75), device=dev) * 0.1).to(torch.bfloat16)
    b = torch.randn(G, 32, device=dev) * 0.1
    for act in ("softplus", "relu", None):
        out["conv1_act_%s" % act] = hot_us(lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (3, 5, 5), 1, 2, 1, act=act))
    out["conv1_pool32_fused_softplus"] = hot_us(lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (3, 5, 5), 1, 2, 1, act="softplus", pool=(3, 2)))
    out["conv1_pool32_fused_none"] = hot_us(lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (3, 5, 5), 1, 2, 1, act=None, pool=(3, 2)))
    for name, (C, H) in {"pool1": (32, 32), "pool2": (64, 15), "pool3": (128, 5)}.items():
        t = torch.rand(G, C, H, H, B, device=dev).to(torch.bfloat16)
        out[name] = hot_us(lambda: ops.maxpool_chwn_bf16(t, 3, 2))
    x2 = torch.rand(G, 32, 15, 15, B, device=dev).to(torch.bfloat16)
    w2 = (torch.randn(G, 64, ops.bf16_row_pitch(800), device=dev) * 0.05).to(torch.bfloat16)
    b2 = torch.randn(G, 64, device=dev) * 0.1
    for act in ("softplus", None):
        out["conv2_act_%s" % act] = hot_us(lambda: ops.conv2d_chwn_bf16_forward(x2, w2, b2, (32, 5, 5), 1, 2, 1, act=act, tap_major=True))
print(json.dumps(out))
