"""Round 4: the ILV threshold (staging loads interleaved with the MFMAs for launches of <= PCONV_ILV_MAX items) was set in round 3
for one step per launch; with four steps per launch conv1 / conv2 / conv3 are 10240 / 7680 / 3840 workgroups.  Variant libraries
built with -DPCONV_ILV_MAX=0 / 4000 / 8000 (scratch/libs, see the notes) against the shipped 12000: ms per step (G = 4, two lanes)
and the six GEMM launches, one subprocess per library (BBB_HIP_LIB), three rounds interleaved."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import json, os, sys, statistics, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "pytorch-bayesiancnn_amd"))
import torch, bench
from bbb_hip import ensemble, rng
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]; net, x = bench.build_net(cfg, dev); E = cfg["E"]; G = 4
with torch.no_grad():
    pipe = ensemble.GraphedPipeline(net, x, E, depth=2, steps_per_launch=G)
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
print("RESULT " + json.dumps({"ms_per_step": round(statistics.median(vals) * 1e3, 4), "per_launch_us": rec.per_launch_us}))
'''
libs = {"12000 (shipped)": None}
for v in (0, 4000, 8000):
    libs[str(v)] = os.path.join(ROOT, "scratch", "libs", "libbbb_ilv%d.so" % v)
for rnd in range(3):
    for tag, lib in libs.items():
        env = dict(os.environ)
        if lib:
            env["BBB_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print(json.dumps({"ilv_max": tag, **(json.loads(line[0][7:]) if line else {"error": p.stderr[-300:]})}), flush=True)
