"""Same-process A/B of ops.wgrad_in_place on the bs 512 x 10 training step (eager)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import zoo, rng, train, ops
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
lt, B, E = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("bbb", 512, 10)
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRIORS, lt, "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(B, 3, 32, 32).cuda(); y = torch.randint(0, 10, (B,)).cuda()
opt = train.FusedAdam(net.parameters(), lr=1e-3)
def run(n):
    for _ in range(n):
        train.train_step(net, opt, x, y, E, 0.1, 50000.0, graph=False)
run(30); torch.cuda.synchronize()
for rnd in range(3):
    for flag in (True, False):
        ops.wgrad_in_place[0] = flag
        run(10); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize()
        print("in_place=%s  %.3f ms per step" % (flag, (time.perf_counter() - t0) * 25))
