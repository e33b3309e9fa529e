"""Round 4: G steps x E draws per launch at the METRIC configuration (AlexNet bs 512, num_ens 10) -- ms per step for
G in {1, 2, 3, 4} x lanes in {1, 2, 3, 4}, the fused reparam + KL pass at 10 / 20 / 30 / 40 draws per launch (plain vs
non-temporal stores are the library's choice), and the six GEMM launches of a step at G * 10 slabs (graph of 10 replays each)."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
from bbb_hip import ensemble, ops, rng
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]
net, x = bench.build_net(cfg, dev)
E = cfg["E"]


def ms_per_step(G, depth, n=240):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=depth, steps_per_launch=G)
        n = -(-n // (G * depth)) * G * depth
        t_end = time.perf_counter() + 0.4                      # pre-heat
        while time.perf_counter() < t_end:
            for _ in range(G * depth):
                pipe.step()
            pipe.sync()
        vals = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            pipe.sync()
            vals.append((time.perf_counter() - t0) / n)
    del pipe
    return round(statistics.median(vals) * 1e3, 4)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "steps"):
    for G in (1, 2, 3, 4):
        row = {"G": G}
        for depth in (1, 2, 3, 4):
            row[f"lanes{depth}"] = ms_per_step(G, depth)
        print(json.dumps(row), flush=True)

if which in ("all", "reparam"):
    mus, rhos, ids = [], [], []
    for l in ensemble.bayesian_layers(net):
        m, r, i = l._param_lists()
        mus += m; rhos += r; ids += i
    n_params = sum(m.numel() for m in mus)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g):
            for _ in range(reps):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e-3 / reps)
        return statistics.median(ts)
    with torch.no_grad():
        for Et in (1, 10, 16, 20, 30, 40):
            t = timed(lambda: ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 1, 0, draws=Et), 10)
            b = (8 + 4 * Et) * n_params
            print(json.dumps({"reparam_draws": Et, "us": round(t * 1e6, 2), "GBps": round(b / t / 1e9, 1), "frac_of_8TBps": round(b / t / 8e12, 4)}), flush=True)

if which in ("all", "gemm"):
    for G in (1, 2, 4):
        rec = bench.LaunchRecorder()
        rec.reps = 10
        xg = x.repeat(G, 1, 1, 1) if G > 1 else x
        with torch.no_grad():
            seed, call0 = rng.next_calls(G * E)
            if G > 1:
                ensemble._local_lse(net, xg, E, seed, call0, E, timers=rec, groups=G)
            else:
                ensemble.mc_forward(net, x, E, timers=rec)
            torch.cuda.synchronize()
            agg = rec.time_in_graphs(dev)
        g = agg["conv_gemm"]
        print(json.dumps({"gemm_G": G, "per_launch_us": rec.per_launch_us, "us_per_step": round(1e3 * g["ms"] / G, 1),
                          "TFLOPs": round(g["work"] / (g["ms"] * 1e-3) / 1e12, 1)}), flush=True)
