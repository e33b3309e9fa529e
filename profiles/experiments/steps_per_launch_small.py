"""Round 5 (review items 4 / 5): the single-draw configurations -- configs[1] (3Conv3FC bf16, bs 256) and configs[2] (AlexNet LRT,
CIFAR-100, bs 512) -- are chains of 10-45 us launches; steps per launch G x lanes swept beyond the G = 4 x 4 lanes that bench.py
times.  ms per step (bench.run_config: hipGraph lanes, pre-heated, 3 blocks -> median)."""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["configs[1]", "configs[2]"]
for name in names:
    c = bench.CONFIGS[name]
    combos = [tuple(int(v) for v in c.split(":")) for c in os.environ["SWEEP"].split(",")] if os.environ.get("SWEEP") else \
        ((4, 4), (8, 4), (8, 2), (16, 2), (16, 4), (32, 2))
    for G, depth in combos:
        vals = []
        try:
            for _ in range(int(os.environ.get("SWEEP_REPS", "3"))):
                nst = G * depth * 6
                r, n2, x2 = bench.run_config(c, nst, G * depth, depth, dev, want_roofline=False, steps_per_launch=G, preheat_s=0.15, single_lane=False)
                vals.append(r["ms_per_step"])
                del n2, x2
                torch.cuda.empty_cache()
            print(json.dumps({"config": name, "G": G, "lanes": depth, "ms_per_step": round(statistics.median(vals), 5), "all": vals}), flush=True)
        except Exception as exc:
            print(json.dumps({"config": name, "G": G, "lanes": depth, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}), flush=True)
