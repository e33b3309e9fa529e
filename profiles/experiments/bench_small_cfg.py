"""Why do the one-draw configurations time worse inside bench.py than in steps_per_launch.py?  run_config on configs[1] in a fresh
process: timed-region length, with / without the earlier sections of a bench run."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
c = bench.CONFIGS["configs[1]"]
what = sys.argv[1] if len(sys.argv) > 1 else "plain"
if what in ("after_head", "after_all"):
    bench.run_config(bench.CONFIGS["metric"], 100, 20, 3, dev, want_roofline=(what == "after_all"))
if what == "after_all":
    bench.dropin_loop(dev, 20)
    bench.training_step(dev, 20)
    torch.cuda.empty_cache()
for nst, spl in ((112, 4), (112, 1), (480, 4), (480, 1), (112, 4)):
    r, _, _ = bench.run_config(c, nst, 5, 4, dev, want_roofline=False, steps_per_launch=spl)
    print(json.dumps({"what": what, "steps": nst, "spl": spl, "ms": r["ms_per_step"]}), flush=True)
