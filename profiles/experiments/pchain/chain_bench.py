"""Persistent chain kernel (csrc/pchain.hip) vs one launch per layer: bitwise equality and step time.

    python profiles/experiments/chain_bench.py [check] [time] [sim]
check: chain == layers (bitwise) on the metric config, E = 1 / 25, 3Conv3FC, work units, graph replay, 3 lanes in flight
time:  ms per step, single lane and 3 lanes, for chain_group in (1, 2, 5, 10) and for per-layer launches
sim:   the busiest rank's share of the 8 / 4 / 2-rank strong-scaling step (rank_sim.py's measurement) with the chain
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
what = set(sys.argv[1:]) or {"check", "time", "sim"}


def build(kind, classes, B, hw=32):
    torch.manual_seed(0)
    net = zoo.getModel(kind, 3, classes, PRI, "bbb", "softplus").to(dev)
    rng.assign_stream_ids(net)
    return net, torch.rand(B, 3, hw, hw, device=dev)


def logits(net, x, E, chain, group=0, units=None):
    ensemble.use_chain, ensemble.chain_flags = chain, group
    with torch.no_grad():
        if units is None:
            out = ensemble._mc_logits_chwn(net, x, E, 7, 3)
        else:
            out = ensemble._mc_logits_chwn(net, x, E, 7, 3, units=units)
    torch.cuda.synchronize()
    return out[0].clone(), out[1].clone(), ensemble.stats["launch"]


if "check" in what:
    res = []
    for kind, classes, B, E in (("alexnet", 10, 512, 10), ("alexnet", 10, 512, 1), ("alexnet", 10, 512, 25), ("alexnet", 100, 256, 3),
                                ("3conv3fc", 10, 256, 2), ("alexnet", 10, 128, 5)):
        net, x = build(kind, classes, B)
        a, kla, la = logits(net, x, E, False)
        for group in (0, 1):
            b, klb, lb = logits(net, x, E, True, group)
            res.append({"case": f"{kind}-{classes} B={B} E={E} group={group}", "launch": [la, lb], "bitwise": bool(torch.equal(a, b)),
                        "kl_equal": bool(torch.equal(kla, klb)), "max_abs_diff": float((a - b).abs().max()),
                        "err_word": ops.chain_error(dev)})
    # work units: 8 ranks -> S = 4; rank 3's share
    net, x = build("alexnet", 10, 512)
    S = ensemble.plan_slices(10, 8, 512)
    for rank in (0, 3, 7):
        lo, hi = ensemble.unit_range(10, S, rank, 8)
        a, _, la = logits(net, x, 10, False, units=(S, lo, hi))
        b, _, lb = logits(net, x, 10, True, 0, units=(S, lo, hi))
        res.append({"case": f"units S={S} rank={rank} [{lo},{hi})", "launch": [la, lb], "bitwise": bool(torch.equal(a, b)),
                    "err_word": ops.chain_error(dev)})
    # graph lanes: 3 steps in flight, chain vs layers, every step compared
    for chain in (False, True):
        ensemble.use_chain, ensemble.chain_flags = chain, 0
        rng.manual_seed(11, 0)
        with torch.no_grad():
            pipe = ensemble.GraphedPipeline(net, x, 10, depth=3)
            outs = []
            for i in range(30):
                lo_, kl_ = pipe.step()
                pipe.sync()
                outs.append(lo_.clone())
        if chain:
            same = all(torch.equal(u, v) for u, v in zip(outs, ref_outs))
            res.append({"case": "GraphedPipeline depth=3, 30 steps, chain vs layers", "bitwise": bool(same),
                        "err_words": [int(ops._scratch[k][8].item()) for k in ops._scratch if k[1] == "chain"]})
        else:
            ref_outs = outs
        del pipe
    # 3 lanes replayed back to back WITHOUT syncing in between (the timed region's mode), last outputs compared
    finals = {}
    for chain in (False, True):
        ensemble.use_chain = chain
        rng.manual_seed(12, 0)
        with torch.no_grad():
            pipe = ensemble.GraphedPipeline(net, x, 10, depth=3)
            for i in range(300):
                pipe.step()
            pipe.sync()
            finals[chain] = [l.lse.clone() for l in pipe.lanes]
        del pipe
    res.append({"case": "3 lanes x 100 unsynchronised replays, final lane outputs", "bitwise": all(torch.equal(u, v) for u, v in zip(finals[False], finals[True])),
                "err_words": [int(ops._scratch[k][8].item()) for k in ops._scratch if k[1] == "chain"]})
    for r in res:
        print(json.dumps(r), flush=True)


def time_steps(net, x, E, lanes, n=200):
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


if "time" in what:
    net, x = build("alexnet", 10, 512)
    for E in (10, 25, 1):
        for chain, group in ((False, 0), (True, 0), (True, 1), (True, 2), (True, 0x300)):
            ensemble.use_chain, ensemble.chain_flags = chain, group
            row = {"E": E, "launch": "chain" if chain else "layers", "flags": hex(group)}
            for lanes in (1, 2, 3):
                row[f"ms_{lanes}lane"] = round(1e3 * time_steps(net, x, E, lanes), 4)
            row["Msamples_s_best"] = round(512 * E / min(v for k, v in row.items() if k.startswith("ms_")) / 1e3, 3)
            print(json.dumps(row), flush=True)

if "sim" in what:
    net, x = build("alexnet", 10, 512)
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

    for chain in (False, True):
        ensemble.use_chain, ensemble.chain_flags = chain, 0
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
                print(json.dumps({"launch": "chain" if chain else "layers", "world": world, "S": S, "lanes": depth,
                                  "ms_per_step_busiest_rank": round(worst * 1e3, 4), "projected_samples_per_s": round(5120 / worst, 0)}),
                      flush=True)
