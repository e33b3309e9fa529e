"""Does the persistent chain launch pay where the per-layer path is launch-bound?  Small steps (E = 1, small nets), hipGraph
replay, 1 and 4 lanes, per-layer launches vs the chain (ensemble.use_chain)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, ops, rng, zoo
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")


def time_steps(net, x, E, lanes, n=400):
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


for kind, classes, B, E in (("3conv3fc", 10, 256, 1), ("3conv3fc", 10, 128, 1), ("lenet", 10, 256, 1), ("alexnet", 10, 128, 1),
                            ("alexnet", 10, 256, 1), ("3conv3fc", 10, 256, 4)):
    torch.manual_seed(0)
    net = zoo.getModel(kind, 3, classes, PRI, "bbb", "softplus").to(dev)
    rng.assign_stream_ids(net)
    x = torch.rand(B, 3, 32, 32, device=dev)
    for chain, flags in ((False, 0), (True, 0), (True, 1)):
        ensemble.use_chain, ensemble.chain_flags = chain, flags
        row = {"net": kind, "B": B, "E": E, "chain": chain, "flags": flags}
        for lanes in (1, 4):
            row[f"ms_{lanes}"] = time_steps(net, x, E, lanes)
        row["launch"] = ensemble.stats["launch"]
        row["err"] = ops.chain_error(dev) if chain else 0
        print(json.dumps(row), flush=True)
