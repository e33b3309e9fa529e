import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import zoo, rng
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRIORS, "bbb", "softplus").cuda()
x = torch.rand(512, 3, 32, 32).cuda()
mode = sys.argv[1] if len(sys.argv) > 1 else "nograd"
def loop(n):
    for _ in range(n):
        out, kl = net(x)
ctx = torch.no_grad() if mode == "nograd" else torch.enable_grad()
with ctx:
    loop(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(200); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, "cpu us/call", (t1 - t0) / 200 * 1e6, "incl drain", (t2 - t0) / 200 * 1e6)
    pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
