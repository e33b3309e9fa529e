"""Where pconv_bf16_strip_kernel's time goes (3Conv3FC conv2, bs 256, 16 steps per launch): the launch with parts of the kernel
compiled out (results are wrong in those variants; only the time is of interest) and without the activation.
Needs BBB_FORCE_VARIANTS="STRIP_NO_MFMA STRIP_NO_BARRIER STRIP_NO_LOADS" profiles/experiments/bf16_shape_sweep_build.sh."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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

    out = {"lib": os.path.basename(os.environ["BBB_HIP_LIB"]), "strip": os.environ.get("BBB_BF16_STRIP", "0")}
    with torch.no_grad():
        x = torch.rand(G, 32, 15, 15, 256, device=dev).to(torch.bfloat16)
        w = (torch.randn(G, 64, 800, device=dev) * 0.03).to(torch.bfloat16)
        b = torch.randn(G, 64, device=dev) * 0.1
        if os.environ.get("BBB_BF16_STRIP") == "c8":
            x = ops.to_c8(x)
        for act in ("softplus", None):
            out["act_%s" % act] = hot_us(lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (32, 5, 5), 1, 2, 1, act=act, tap_major=True))
            if x.dim() == 6:
                out["c8out_act_%s" % act] = hot_us(lambda: ops.conv2d_chwn_bf16_forward(x, w, b, (32, 5, 5), 1, 2, 1, act=act, tap_major=True, out_c8=True))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("BF16_STRIP_CHILD"):
        child(16)
        sys.exit(0)
    strips = sys.argv[1:] or ["32", "42"]
    if strips[0] == "smem":
        for n in (0, 50000):
            env = dict(os.environ, BF16_STRIP_CHILD="1", BBB_BF16_STRIP="c8", BBB_STRIP_SMEM_PAD=str(n), BBB_HIP_LIB=os.path.join(ROOT, "build_var", "libbbb_force.so"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=200)
            print(n, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-300:], flush=True)
        sys.exit(0)
    if strips[0] == "stagger":
        for n in (0, 6, 256 + 6, 512 + 6, 768 + 6, 256 + 12, 512 + 12):
            env = dict(os.environ, BF16_STRIP_CHILD="1", BBB_BF16_STRIP="c8", BBB_STRIP_STAGGER=str(n), BBB_HIP_LIB=os.path.join(ROOT, "build_var", "libbbb_force.so"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=200)
            print(n, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-300:], flush=True)
        sys.exit(0)
    for lib in ("", "STRIP_NO_MFMA", "STRIP_NO_LOADS", "STRIP_NO_EPILOGUE", "STRIP_NO_STORE"):
        for v in strips:
            path = os.path.join(ROOT, "build_var", "libbbb_force%s.so" % lib)
            if not os.path.exists(path):
                continue
            env = dict(os.environ, BF16_STRIP_CHILD="1", BBB_BF16_STRIP=v, BBB_HIP_LIB=path)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=200)
            print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else json.dumps({"lib": lib, "error": r.stderr[-400:]}), flush=True)
