"""Round 5, review item 2: resident workgroups that claim their items from per-XCD counters (csrc/pconv_gemm.hip run_items) against
one workgroup per item, per launch class (apply resident_wgs.patch to csrc/ first: the experiment is NOT in the shipped library; (BBB_PCONV_RESIDENT bit mask: 1 plain, 2 pooled conv1, 4 in-workgroup split = conv4 / conv5)).
Metric step, G = 4 x 2 lanes: ms per step, the six GEMM launches, and a digest of the step's output (must not move: which workgroup
computes an item does not change the item).  One subprocess per mask, rounds interleaved."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
WORKER = r'''
import hashlib, json, os, sys, statistics, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ensemble, rng
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]; net, x = bench.build_net(cfg, dev); E = cfg["E"]; G = 4
with torch.no_grad():
    rng.manual_seed(1234)
    pipe = ensemble.GraphedPipeline(net, x, E, depth=2, steps_per_launch=G)
    outs = []
    for _ in range(8):
        lo, kl = pipe.step()
        outs.append((lo, kl))
    pipe.sync()
    h = hashlib.sha256()
    for lo, kl in outs[-4:]:
        h.update(lo.cpu().numpy().tobytes()); h.update(kl.cpu().numpy().tobytes())
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(8): pipe.step()
        pipe.sync()
    vals = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(240): pipe.step()
        pipe.sync(); vals.append((time.perf_counter() - t0) / 240)
    del pipe
    rec = bench.LaunchRecorder(); rec.reps = 10
    xg = x.repeat(G, 1, 1, 1)
    seed, call0 = rng.next_calls(G * E)
    ensemble._local_lse(net, xg, E, seed, call0, E, timers=rec, groups=G)
    torch.cuda.synchronize(); rec.time_in_graphs(dev)
print("RESULT " + json.dumps({"ms_per_step": round(statistics.median(vals) * 1e3, 4), "min": round(min(vals) * 1e3, 4),
                              "per_launch_us": rec.per_launch_us, "digest": h.hexdigest()[:16]}))
'''
masks = [int(m) for m in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,4,3".split(","))]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for rnd in range(rounds):
    for m in masks:
        env = dict(os.environ, BBB_PCONV_RESIDENT=str(m))
        p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print(json.dumps({"resident_mask": m, **(json.loads(line[0][7:]) if line else {"error": p.stderr[-400:]})}), flush=True)
