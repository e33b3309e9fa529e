"""Round 4: what a short timed region costs.  Blocks of N steps between device syncs (the bench's timed_block), N = 4 .. 200, median
of 7 each, G = 4 x 2 lanes: T(N) = a + b N; plus the host time of one graph launch and the same blocks with the host spinning on
an event instead of sleeping in hipDeviceSynchronize (ends the block a wake-up latency earlier)."""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ensemble
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]; net, x = bench.build_net(cfg, dev); E = cfg["E"]
with torch.no_grad():
    pipe = ensemble.GraphedPipeline(net, x, E, depth=2, steps_per_launch=4)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for _ in range(8): pipe.step()
        pipe.sync()

    def block(n, spin=False, idle=0.0):
        torch.cuda.synchronize(dev)
        if idle:
            time.sleep(idle)
        t0 = time.perf_counter()
        for _ in range(n): pipe.step()
        pipe.i = -(-pipe.i // pipe.G) * pipe.G
        if spin:
            evs = []
            for lane in pipe.lanes:
                ev = torch.cuda.Event(); ev.record(lane.stream); evs.append(ev)
            while not all(e.query() for e in evs):
                pass
        else:
            for lane in pipe.lanes: lane.stream.synchronize()
            torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    rows = []
    for n in (4, 8, 20, 40, 100, 200):
        ts = [block(n) for _ in range(7)]
        tsp = [block(n, spin=True) for _ in range(7)]
        rows.append((n, statistics.median(ts), statistics.median(tsp)))
        print(json.dumps({"steps": n, "ms_block": round(rows[-1][1] * 1e3, 4), "ms_per_step": round(rows[-1][1] / n * 1e3, 4),
                          "spin_wait_ms_per_step": round(rows[-1][2] / n * 1e3, 4)}), flush=True)
    (n1, t1, _), (n2, t2, _) = rows[2], rows[-1]
    b = (t2 - t1) / (n2 - n1); a = t1 - b * n1
    print(json.dumps({"fit_20_200": {"fixed_ms_per_block": round(a * 1e3, 4), "ms_per_step": round(b * 1e3, 4)}}))
    for idle in (0.0, 0.001, 0.01, 0.1):
        ts = [block(20, idle=idle) for _ in range(7)]
        print(json.dumps({"steps": 20, "host_idle_before_block_s": idle, "ms_per_step": round(statistics.median(ts) / 20 * 1e3, 4)}), flush=True)
    # host cost of one graph launch
    lane = pipe.lanes[0]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.cuda.stream(lane.stream):
        lane.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    print(json.dumps({"host_us_one_graph_launch": round((t1 - t0) * 1e6, 1)}))
