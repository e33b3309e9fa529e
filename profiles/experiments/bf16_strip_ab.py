"""The strip form (a strip of output pixels per workgroup; "c8" = pconv_bf16_strip8_kernel over channel-interleaved input) against the
general bf16 kernel on 3Conv3FC conv2 (bs 256, G steps per launch): time per launch and a hash of the output bytes (the strip form
promises the general kernel's one-k-group bits: equal hashes from G = 4 on), plus ragged shapes (hash only; "+" = the
channel-interleaved OUTPUT holds the same tensor).  Runs on the shipped library.     usage: bf16_strip_ab.py [G]"""
import hashlib, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [0, "c8"]


def child(G):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
    import torch, bench
    from bbb_hip import ops
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
        return round(statistics.median(ts) * 1e3, 2)

    def case(E, Ci, H, W, Co, k, pad, B, act, time_it=False):
        torch.manual_seed(E * 1000 + H * 10 + B)
        K = Ci * k * k
        x = torch.rand(E, Ci, H, W, B, device=dev).to(torch.bfloat16)
        w = torch.zeros(E, Co, ops.bf16_row_pitch(K), device=dev)
        w[:, :, :K] = torch.randn(E, Co, K, device=dev) * (1.0 / K ** 0.5)
        w = w.to(torch.bfloat16)
        b = torch.randn(E, Co, device=dev) * 0.1
        xin = ops.to_c8(x) if os.environ.get("BF16_STRIP_C8") else x
        f = lambda: ops.conv2d_chwn_bf16_forward(xin, w, b, (Ci, k, k), 1, pad, 1, act=act, tap_major=True)
        y = f()
        torch.cuda.synchronize()
        h = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        if xin.dim() == 6 and Co % 8 == 0:      # the channel-interleaved output must be the same tensor
            y8 = ops.from_c8(ops.conv2d_chwn_bf16_forward(xin, w, b, (Ci, k, k), 1, pad, 1, act=act, tap_major=True, out_c8=True))
            h += "+" if torch.equal(y8, y) else "-MISMATCH"
        return (h, hot_us(f)) if time_it else (h, None)

    out = {"strip": "c8" if os.environ.get("BF16_STRIP_C8") else int(os.environ.get("BBB_BF16_STRIP", "0")), "G": G}
    with torch.no_grad():
        out["conv2_hash"], out["conv2_us"] = case(G, 32, 15, 15, 64, 5, 2, 256, "softplus", True)
        # ragged / odd shapes, hash only: (E, Cin, H, W, Cout, k, pad, B, act)
        for name, c in {"w13_b200_co48": (40, 32, 9, 13, 48, 5, 2, 200, "relu"), "pad0_w16": (48, 32, 12, 16, 64, 5, 0, 128, None),
                        "pad1_co100_b136": (24, 32, 8, 11, 100, 5, 1, 136, "softplus"), "h5w5": (64, 32, 5, 5, 64, 5, 2, 256, "softplus"),
                        "pad4_w9": (64, 32, 9, 9, 64, 5, 4, 128, "relu")}.items():
            out[name] = case(*c)[0]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("BF16_STRIP_CHILD"):
        child(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
        sys.exit(0)
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for v in VARIANTS:
        env = dict(os.environ, BF16_STRIP_CHILD="1", BBB_BF16_STRIP="0" if v == "c8" else str(v))
        if v == "c8":
            env["BF16_STRIP_C8"] = "1"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(G)], env=env, capture_output=True, text=True, timeout=200)
            ok = r.returncode == 0 and r.stdout.strip()
            print(r.stdout.strip().splitlines()[-1] if ok else json.dumps({"strip": v, "error": r.stderr[-400:]}), flush=True)
        except subprocess.TimeoutExpired:
            print(json.dumps({"strip": v, "error": "timeout"}), flush=True)
