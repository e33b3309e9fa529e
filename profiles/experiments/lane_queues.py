"""How many steps in flight pay, and on which streams?  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4)
round-robin; graph branches take queue slots too.  E x lanes x {pooled lane streams, fresh streams per pipeline}, run under
different GPU_MAX_HW_QUEUES from the shell."""
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
pooled = ensemble._lane_streams


def time_steps(E, lanes, n=300):
    with torch.no_grad():
        pipe = ensemble.GraphedPipeline(net, x, E, depth=lanes)
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


for rep in range(2):
    for E in (5, 1):
        for mode in ("pooled", "fresh"):
            ensemble._lane_streams = pooled if mode == "pooled" else (lambda d, n: [torch.cuda.Stream(device=d) for _ in range(n)])
            row = {"queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "E": E, "streams": mode}
            for lanes in (1, 2, 3, 4):
                row[f"ms_{lanes}"] = time_steps(E, lanes)
            print(json.dumps(row), flush=True)
