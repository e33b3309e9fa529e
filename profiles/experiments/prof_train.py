import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'pytorch-bayesiancnn_amd'))
import torch
from bbb_hip import zoo, rng, train
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
lt = sys.argv[1] if len(sys.argv) > 1 else "lrt"
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, PRIORS, lt, "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(256, 3, 32, 32).cuda(); y = torch.randint(0, 10, (256,)).cuda()
opt = train.FusedAdam(net.parameters(), lr=1e-3)
def loop(n):
    for _ in range(n): train.train_step(net, opt, x, y, 1, 0.1, 50000.0)
loop(10); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(100); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(lt, "host us/step", (t1 - t0) / 100 * 1e6, "incl drain", (t2 - t0) / 100 * 1e6)
pr = cProfile.Profile(); pr.enable(); loop(100); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:7000])
