"""15 eager training steps (forward + backward + Adam) of BayesianAlexNet: usage train_steps.py <bbb|lrt> <batch> <draws>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import zoo, rng, train
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
lt, B, E = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRIORS, lt, "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(B, 3, 32, 32).cuda(); y = torch.randint(0, 10, (B,)).cuda()
opt = train.FusedAdam(net.parameters(), lr=1e-3)
for _ in range(5):
    train.train_step(net, opt, x, y, E, 0.1, 50000.0, graph=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    train.train_step(net, opt, x, y, E, 0.1, 50000.0, graph=False)
torch.cuda.synchronize()
print(lt, B, E, "ms per step %.3f" % ((time.perf_counter() - t0) * 100))
if os.environ.get("TRAIN_STEPS_LONG", "1") == "1":      # a longer, warmer measurement (the first ten steps above are what the trace script cuts)
    for _ in range(30):
        train.train_step(net, opt, x, y, E, 0.1, 50000.0, graph=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        train.train_step(net, opt, x, y, E, 0.1, 50000.0, graph=False)
    torch.cuda.synchronize()
    print(lt, B, E, "warm ms per step %.3f" % ((time.perf_counter() - t0) * 25))
