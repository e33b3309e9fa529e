"""Why does splitting ONE small layer cost throughput with several steps in flight?  E = 5 AlexNet bs 512: split off / on / forced
per layer, 1 and 3 lanes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, ops, rng, zoo
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRI, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device=dev)
orig = ops._split_scratch


def time_steps(E, lanes, n=300):
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
    return round(best * 1e3, 4)


for E in (5, 5, 6, 6, 5, 1, 1):
    for mode in ("off", "on", "on_single_wg_combine_off"):
        if mode == "off":
            ops.split_k = False
        else:
            ops.split_k = True
        ops._split_plans.clear()
        row = {"E": E, "mode": mode}
        for lanes in (1, 2, 3):
            row[f"ms_{lanes}"] = time_steps(E, lanes)
        print(json.dumps(row), flush=True)
        if mode == "on":
            break
