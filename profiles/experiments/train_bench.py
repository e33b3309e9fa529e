import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, zoo, rng, train
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
def run(net_type, lt, B, E, fast, graphed):
    torch.manual_seed(0)
    net = zoo.getModel(net_type, 3, 10, PRI, lt, "softplus").to(dev)
    rng.assign_stream_ids(net)
    x = torch.rand(B, 3, 32, 32, device=dev); y = torch.randint(0, 10, (B,), device=dev)
    ensemble.fast_autograd = fast
    opt = train.FusedAdam(net.parameters(), lr=1e-3, capturable=graphed)
    if graphed:
        g = train.GraphedTrainStep(net, opt, x, y, E, 0.1, 50000.0, warmup=3)
        step = g.step
    else:
        step = lambda: train.train_step(net, opt, x, y, E, 0.1, 50000.0)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize()
    ensemble.fast_autograd = True
    return round((time.perf_counter() - t0) / n * 1e3, 3), ensemble.stats["path"]
for cfg in [("alexnet", "bbb", 512, 10), ("alexnet", "bbb", 256, 1), ("alexnet", "lrt", 256, 1), ("alexnet", "lrt", 512, 10), ("3conv3fc", "lrt", 256, 1)]:
    for fast in (True, False):
        for graphed in (False, True):
            try:
                ms, path = run(*cfg, fast, graphed)
                print(json.dumps({"cfg": cfg, "fast": fast, "graphed": graphed, "ms_per_step": ms, "path": path}), flush=True)
            except Exception as e:
                print(json.dumps({"cfg": cfg, "fast": fast, "graphed": graphed, "error": repr(e)[:300]}), flush=True)
