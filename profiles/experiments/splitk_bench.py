"""Split contraction for small launches (bbb_conv2d_chwn_splitk_fwd): on vs off, same process.

    python profiles/experiments/splitk_bench.py [step] [layers] [sim]
step:   ms per Monte-Carlo step (hipGraph, 1 and 3 lanes) for E in (1, 2, 3, 5), BBB and LRT AlexNet bs 512, and the largest
        relative difference of the logits between the two modes
layers: us per AlexNet layer at E = 1 (20 launches per graph), BBB and LRT kernels
sim:    the busiest rank's share of the 8 / 4-rank strong-scaling step (rank_sim.py's measurement), on vs off
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch  # noqa: E402
from bbb_hip import ensemble, ops, rng, zoo  # noqa: E402

PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
what = set(sys.argv[1:]) or {"step", "layers", "sim"}


def build(lt, classes=10, B=512, kind="alexnet"):
    torch.manual_seed(0)
    net = zoo.getModel(kind, 3, classes, PRI, lt, "softplus").to(dev)
    rng.assign_stream_ids(net)
    return net, torch.rand(B, 3, 32, 32, device=dev)


def time_steps(net, x, E, lanes, n=300):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=lanes) if lanes > 1 else ensemble.GraphedMC(net, x, E)
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
    return best


if "step" in what:
    for lt, classes in (("bbb", 10), ("lrt", 100)):
        net, x = build(lt, classes)
        for E in (1, 2, 3, 5):
            row = {"layer_type": lt, "E": E}
            outs = {}
            for on in (False, True):
                ops.split_k = on
                with torch.no_grad():
                    lg, kl = ensemble._mc_logits_chwn(net, x, E, 7, 3)
                outs[on] = lg.clone()
                for lanes in (1, 3):
                    row[f"ms_{lanes}lane_{'split' if on else 'plain'}"] = round(1e3 * time_steps(net, x, E, lanes), 4)
            a, b = outs[False], outs[True]
            row["max_rel_diff"] = float((a - b).abs().max() / a.abs().max())
            row["run_to_run_bitwise"] = bool(torch.equal(b, ensemble._mc_logits_chwn(net, x, E, 7, 3)[0]))
            print(json.dumps(row), flush=True)

if "layers" in what:
    L = [("conv1", 3, 32, 32, 64, 11, 4, 5), ("conv2", 64, 4, 4, 192, 5, 1, 2), ("conv3", 192, 2, 2, 384, 3, 1, 1),
         ("conv4", 384, 2, 2, 256, 3, 1, 1), ("conv5", 256, 2, 2, 128, 3, 1, 1), ("fc", 128, 1, 1, 10, 1, 1, 0)]

    def t(fn, n=20, reps=5):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 1e3 / n)
        return best
    for E in (1, 2):
        for lrt in (False, True):
            row = {"E": E, "kernel": "lrt" if lrt else "bbb"}
            for name, Cin, H, W, Cout, k, st, pd in L:
                torch.manual_seed(0)
                x = torch.randn(E, Cin, H, W, 512, device=dev)
                w = torch.randn(1 if lrt else E, Cout, Cin, k, k, device=dev) * 0.05
                b = torch.randn(1 if lrt else E, Cout, device=dev)
                r = []
                for on in (False, True):
                    ops.split_k = on
                    if lrt:
                        w2 = w[0].abs() * 0.01
                        fn = lambda: ops.lrt_conv2d_chwn_forward(x, w[0], w2, b[0], b[0].abs(), 1, 0, 2, st, pd, 1, act="softplus")
                    else:
                        fn = lambda: ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
                    r.append(round(t(fn), 1))
                row[name] = r
            row["total_plain_split"] = [round(sum(row[n[0]][i] for n in L), 1) for i in (0, 1)]
            print(json.dumps(row), flush=True)

if "sim" in what:
    net, x = build("bbb")
    E = 10

    class Lane:
        def __init__(self, S, lo, hi, lane, lanes):
            self.counter = torch.full((1,), lane * E, dtype=torch.int32, device=dev)
            self.stream = torch.cuda.Stream()
            self.S, self.lo, self.hi = S, lo, hi
            self.stride = lanes * E
            with torch.no_grad(), torch.cuda.stream(self.stream), rng.device_call_offset(self.counter):
                for _ in range(2):
                    self.body()
            torch.cuda.synchronize()
            self.g = torch.cuda.CUDAGraph()
            with torch.no_grad(), rng.device_call_offset(self.counter), torch.cuda.graph(self.g, stream=self.stream, capture_error_mode="thread_local"):
                self.out = self.body()

        def body(self):
            if self.S > 1:
                lse, kl = ensemble._local_lse(net, x, E, 1, 0, 0, units=(self.S, self.lo, self.hi))
            else:
                lse, kl = ensemble._local_lse(net, x, self.hi - self.lo, 1, self.lo, 0)
            self.counter.add_(self.stride)
            return lse, kl

        def step(self):
            with torch.cuda.stream(self.stream):
                self.g.replay()

    for on in (False, True):
        ops.split_k = on
        for world in (1, 2, 4, 8):
            S = ensemble.plan_slices(E, world, 512)
            for depth in (1, 3, 4):
                worst = 0
                for rank in sorted({0, world - 1}):
                    lo, hi = ensemble.unit_range(E, S, rank, world)
                    lanes = [Lane(S, lo, hi, l, depth) for l in range(depth)]
                    for i in range(30):
                        lanes[i % depth].step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = 300
                    for i in range(n):
                        lanes[i % depth].step()
                    torch.cuda.synchronize()
                    worst = max(worst, (time.perf_counter() - t0) / n)
                    del lanes
                print(json.dumps({"split_k": on, "world": world, "S": S, "lanes": depth, "ms_per_step_busiest_rank": round(worst * 1e3, 4),
                                  "projected_samples_per_s": round(5120 / worst, 0)}), flush=True)
