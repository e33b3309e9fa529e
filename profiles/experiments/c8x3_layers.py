"""AlexNet CIFAR conv2 .. conv5 + classifier at bs 512, E slabs per launch: the fp32 MFMA kernel, round 4's split-bf16 kernel (S3 in,
S3 out) and round 6's split-bf16 kernel over MFMA-ready operands (c8 S3 + tap-major weights), hot (hipGraph of 10 launches,
pre-heated), us per launch and fraction of the 16-bit matrix peak.  usage: c8x3_layers.py [E ...]   (env BBB_C8X3_MT=1|2 forces a tile)"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ops
from bbb_hip.ensemble import conv_flops

dev = torch.device("cuda:0"); torch.cuda.set_device(0)


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
    return statistics.median(ts) * 1e3


LAYERS = {"conv2": ((64, 4, 4), (192, 64, 5, 5), 2), "conv3": ((192, 2, 2), (384, 192, 3, 3), 1),
          "conv4": ((384, 2, 2), (256, 384, 3, 3), 1), "conv5": ((256, 2, 2), (128, 256, 3, 3), 1),
          "fc": ((128, 1, 1), (10, 128, 1, 1), 0)}
B = 512
for E in [int(a) for a in sys.argv[1:]] or [40, 10]:
    rows = {}
    with torch.no_grad():
        for name, ((C, H, W), (Co, Ci, kh, kw), pad) in LAYERS.items():
            torch.manual_seed(0)
            x = torch.rand(E, C, H, W, B, device=dev)
            w = torch.randn(E, Co, Ci, kh, kw, device=dev) * (1.0 / (Ci * kh * kw) ** 0.5)
            b = torch.randn(E, Co, device=dev) * 0.1
            fl = conv_flops(B, C, H, W, Co, kh, kw, 1, pad, 1, E)[0]
            of32 = name == "fc"
            xs, xc, wt = ops.s3_from_f32(x), ops.c8s3_from_f32(x), ops.w_tap_major(w)
            f32 = lambda: ops.conv2d_chwn_forward(x, w, b, 1, pad, 1, act="softplus", bf16x3=False)
            old = lambda: ops.conv2d_chwn_forward(xs, w, b, 1, pad, 1, act="softplus", bf16x3=True, x_s3=True, out_s3=not of32)
            new = lambda: ops.conv2d_c8x3_forward(xc, wt, b, (kh, kw), 1, pad, 1, act="softplus", out_f32=of32)
            ref = f32()
            got = new() if of32 else ops.c8s3_to_f32(new())
            err = float((got - ref).abs().max() / ref.abs().max())
            quick = os.environ.get("C8X3_ONLY") == "1"
            t32, told, tnew = (0.0, 0.0, hot_us(new)) if quick else (hot_us(f32), hot_us(old), hot_us(new))
            rows[name] = dict(fp32_us=round(t32, 1), bf16x3_us=round(told, 1), c8x3_us=round(tnew, 1), rel_diff_vs_fp32=float(f"{err:.2e}"))
            if os.environ.get("C8X3_SWEEP") == "1" and not of32:
                sw = {}
                for nt in (2, 3, 4):
                    for tile in (128, 256):
                        fn = lambda nt=nt, tile=tile: ops.conv2d_c8x3_forward(xc, wt, b, (kh, kw), 1, pad, 1, act="softplus", nt=nt, tile=tile)
                        sw[f"nt{nt}_t{tile}"] = round(hot_us(fn), 1)
                rows[name]["sweep_us"] = sw
            if fl:
                rows[name]["c8x3_frac_of_bf16_peak"] = round(6 * fl / (tnew * 1e-6) / 2.5e15, 3)
                rows[name]["fp32_frac"] = round(fl / (t32 * 1e-6) / 157.3e12, 3) if t32 else None
    tot = {k: round(sum(r[k] for r in rows.values()), 1) for k in ("fp32_us", "bf16x3_us", "c8x3_us")}
    print(json.dumps({"E": E, "mt": os.environ.get("BBB_C8X3_MT", "auto"), "layers": rows, "total": tot}), flush=True)
