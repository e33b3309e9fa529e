"""Kernel list of ONE rank's share of the 8-rank strong-scaling step (units path), eager, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, rng, zoo
PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRI, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device=dev)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = ensemble.plan_slices(10, world, 512)
lo, hi = ensemble.unit_range(10, S, 3 % world, world)
with torch.no_grad():
    for _ in range(20):
        ensemble._local_lse(net, x, 10, 1, 0, 0, units=(S, lo, hi)) if S > 1 else ensemble._local_lse(net, x, hi - lo, 1, lo, 0)
    torch.cuda.synchronize()
print("S", S, "units", lo, hi)
