"""The reference's default training configuration (lrt, bs 256, 1 draw) as train_step runs it by default (one hipGraph): 40 replays for a
kernel trace.  usage: rocprofv3 --kernel-trace -- python train_default_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import zoo, rng, train
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRIORS, "lrt", "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(256, 3, 32, 32).cuda(); y = torch.randint(0, 10, (256,)).cuda()
opt = train.FusedAdam(net.parameters(), lr=1e-3)
for _ in range(8):
    train.train_step(net, opt, x, y, 1, 0.1, 50000.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    train.train_step(net, opt, x, y, 1, 0.1, 50000.0)
torch.cuda.synchronize()
print("ms per step %.4f" % ((time.perf_counter() - t0) * 25))
