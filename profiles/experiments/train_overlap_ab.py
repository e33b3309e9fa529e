"""Training step (bs 512 x 10, BBB): weight gradients on a side stream beside the input gradients (fast_train.overlap_wgrad) A/B,
eager and as one hipGraph."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import zoo, rng, train, fast_train
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
out = {}
for rnd in (1, 2):
    for flag in (True, False):
        fast_train.overlap_wgrad[0] = flag
        torch.manual_seed(0)
        net = zoo.getModel("alexnet", 3, 10, PRIORS, "bbb", "softplus").cuda()
        rng.assign_stream_ids(net)
        x = torch.rand(512, 3, 32, 32).cuda(); y = torch.randint(0, 10, (512,)).cuda()
        for graph in (False, None):
            opt = train.FusedAdam(net.parameters(), lr=1e-3)
            for _ in range(6):
                train.train_step(net, opt, x, y, 10, 0.1, 50000.0, graph=graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                train.train_step(net, opt, x, y, 10, 0.1, 50000.0, graph=graph)
            torch.cuda.synchronize()
            out[f"round{rnd}_overlap={flag}_graph={graph}"] = round((time.perf_counter() - t0) * 50, 3)
print(json.dumps(out))
