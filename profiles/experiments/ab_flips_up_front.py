"""Same-process A/B of fast_train.flips_up_front: bbb 512 x 10 and lrt 512 x 10 eager, lrt 256 x 1 as one hipGraph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import zoo, rng, train, fast_train
PRIORS = {'prior_mu': 0, 'prior_sigma': 0.1, 'posterior_mu_initial': (0, 0.1), 'posterior_rho_initial': (-5, 0.1)}
for lt, B, E, graph in (("bbb", 512, 10, False), ("lrt", 512, 10, False), ("lrt", 256, 1, True)):
    for rnd in range(2):
        for flag in (True, False):
            fast_train.flips_up_front[0] = flag
            torch.manual_seed(0)
            net = zoo.getModel("alexnet", 3, 10, PRIORS, lt, "softplus").cuda()
            rng.assign_stream_ids(net)
            x = torch.rand(B, 3, 32, 32).cuda(); y = torch.randint(0, 10, (B,)).cuda()
            opt = train.FusedAdam(net.parameters(), lr=1e-3)
            def run(n):
                for _ in range(n):
                    train.train_step(net, opt, x, y, E, 0.1, 50000.0, graph=(None if graph else False))
            run(20); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(60); torch.cuda.synchronize()
            print(lt, B, E, "up_front=%s  %.4f ms per step" % (flag, (time.perf_counter() - t0) * 1e3 / 60))
fast_train.flips_up_front[0] = True
