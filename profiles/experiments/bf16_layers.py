"""bf16 path, per-launch times of the 3Conv3FC bs 256 E = 1 step (BASELINE configs[1]) and of AlexNet bs 512 E = 1: every conv /
linear launch replayed 20x inside a hipGraph; then ms per step, 1 and 4 lanes.  (`ops.split_k` only concerns fp32 launches in the
shipped library; the bf16 split this script was written to measure is described in profiles/r03_notes.md section 8.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble, ops, rng, zoo
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def per_launch(net, x, E, per_draw=False):
    rec = bench.LaunchRecorder()
    with torch.no_grad():
        if per_draw:          # E one-draw steps on E batches in one set of launches (GraphedPipeline steps_per_launch)
            ensemble._mc_logits_chwn(net, x.repeat(E, 1, 1, 1), 1, 7, 3, timers=rec, precision="bf16", groups=E)
        else:
            ensemble._mc_logits_chwn(net, x, E, 7, 3, timers=rec, precision="bf16")
    torch.cuda.synchronize()
    out = []
    st = torch.cuda.Stream()
    for name, flop, fn in rec.calls:
        if name != "conv_gemm":
            continue
        with torch.no_grad(), torch.cuda.stream(st):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                for _ in range(20):
                    fn()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                g.replay()
                e1.record(st)
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        out.append(round(sorted(ts)[2], 2))
    return out


def time_steps(net, x, E, lanes, n=400):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=lanes, precision="bf16")
        for _ in range(30):
            pipe.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
    del pipe
    return round(best * 1e3, 4)


cfg = dict(bench.CONFIGS["configs[1]"])
net, x = bench.build_net(cfg, dev)
for G in (1, 2, 4, 8):
    print(json.dumps({"config": "configs[1]", "steps_per_launch": G, "us_per_launch": per_launch(net, x, G, per_draw=True)}), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "short":
    sys.exit(0)
for name in ("configs[1]", "metric"):
    cfg = dict(bench.CONFIGS[name])
    net, x = bench.build_net(cfg, dev)
    for E in ((1,) if name == "configs[1]" else (1, 10)):
        for on in (False, True):
            ops.split_k = on
            ops._split_plans.clear()
            row = {"config": name, "E": E, "split": on, "us_per_launch": per_launch(net, x, E)}
            row["ms_1"], row["ms_4"] = time_steps(net, x, E, 1), time_steps(net, x, E, 4)
            print(json.dumps(row), flush=True)
