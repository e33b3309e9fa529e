import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
for name, combos in (("configs[3]", ((1, 3), (2, 2), (2, 3), (4, 2))), ("configs[4]", ((1, 3), (2, 2), (1, 2)))):
    c = bench.CONFIGS[name]
    for G, depth in combos:
        vals = []
        try:
            for _ in range(3):
                nst = max(12, G * depth * 4)
                r, n2, x2 = bench.run_config(c, nst, G * depth, depth, dev, want_roofline=False, steps_per_launch=G, preheat_s=0.2, single_lane=False)
                vals.append(r["ms_per_step"]); del n2, x2; torch.cuda.empty_cache()
            print(json.dumps({"config": name, "G": G, "lanes": depth, "ms_per_step": round(statistics.median(vals), 4), "all": vals}), flush=True)
        except Exception as exc:
            print(json.dumps({"config": name, "G": G, "lanes": depth, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}), flush=True)
