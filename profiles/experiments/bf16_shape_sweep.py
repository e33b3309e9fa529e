"""configs[1] (3Conv3FC, bf16, bs 256): conv2 / conv3 / fc1 / fc2 of the general bf16 kernel under every (tile shape, k-groups, wave
specialisation) instantiation the library has, at G steps per launch -- is the launcher's LDS-cycle model picking the fastest one?
Needs a library built from a copy of pconv_bf16.hip that reads BBB_BF16_FORCE (shape*100 + kgs*10 + ws) in the launcher
(build_var/libbbb_force.so, made by profiles/experiments/bf16_shape_sweep_build.sh); the shipped library has no such switch.
usage: bf16_shape_sweep.py [G ...]     (parent: spawns one child per variant)"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [0, 1410, 1420, 1401, 1210, 1220, 1240, 1201, 2210, 2220, 2201]


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

    out = {"force": int(os.environ.get("BBB_BF16_FORCE", "0")), "G": G}
    torch.manual_seed(0)
    B = 256
    layers = {"conv2": ((32, 15, 15), (64, 32, 5, 5), 2), "conv3": ((64, 7, 7), (128, 64, 5, 5), 1),
              "fc1": ((512, 1, 1), (1000, 512, 1, 1), 0), "fc2": ((1000, 1, 1), (1000, 1000, 1, 1), 0)}
    with torch.no_grad():
        for name, ((C, H, W), (Co, Ci, kh, kw), pad) in layers.items():
            K = Ci * kh * kw
            x = torch.rand(G, C, H, W, B, device=dev).to(torch.bfloat16)
            w = torch.zeros(G, Co, ops.bf16_row_pitch(K), device=dev)
            w[:, :, :K] = torch.randn(G, Co, K, device=dev) * (1.0 / K ** 0.5)
            w = w.to(torch.bfloat16)
            b = torch.randn(G, Co, device=dev) * 0.1
            tapm = ops.bf16_tap_major((Co, Ci, kh, kw))
            f = lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (Ci, kh, kw), 1, pad, 1, act="softplus", tap_major=tapm)
            y = f().float()
            out[name] = hot_us(f)
            out[name + "_sum"] = round(float(y.double().abs().sum()), 1)      # same operands in every child: sums agree to rounding
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("BF16_SWEEP_CHILD"):
        child(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
        sys.exit(0)
    for G in ([int(a) for a in sys.argv[1:]] or [16]):
        for v in VARIANTS:
            env = dict(os.environ, BF16_SWEEP_CHILD="1", BBB_BF16_FORCE=str(v), BBB_HIP_LIB=os.path.join(ROOT, "build_var", "libbbb_force.so"))
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), str(G)], env=env, capture_output=True, text=True, timeout=240)
                ok = r.returncode == 0 and r.stdout.strip()
                print(r.stdout.strip().splitlines()[-1] if ok else json.dumps({"force": v, "G": G, "error": r.stderr[-300:]}), flush=True)
            except subprocess.TimeoutExpired:
                print(json.dumps({"force": v, "G": G, "error": "timeout"}), flush=True)
