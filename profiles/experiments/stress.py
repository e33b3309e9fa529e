"""Race hunting for the cross-workgroup protocols (split-contraction tickets): the same launches repeated many
times, alone and with three graph lanes keeping the chip under uneven load, every result compared bitwise with the first."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch  # noqa: E402
from bbb_hip import ensemble, ops, rng, zoo  # noqa: E402

PRI = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
out = {}
for lt, classes in (("bbb", 10), ("lrt", 100)):
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, classes, PRI, lt, "softplus").to(dev)
    rng.assign_stream_ids(net)
    x = torch.rand(512, 3, 32, 32, device=dev)
    big, xb = zoo.getModel("alexnet", 3, 10, PRI, "bbb", "softplus").to(dev), torch.rand(512, 3, 32, 32, device=dev)
    rng.assign_stream_ids(big)
    with torch.no_grad():
        load = ensemble.GraphedPipeline(big, xb, 10, depth=3)          # background load on three other streams
        ref = {E: ensemble._mc_logits_chwn(net, x, E, 7, 3)[0].clone() for E in (1, 2, 3)}
        bad = 0
        t0 = time.time()
        for it in range(400):
            if it % 2:
                for _ in range(3):
                    load.step()
            for E in (1, 2, 3):
                got = ensemble._mc_logits_chwn(net, x, E, 7, 3)[0]
                bad += int(not torch.equal(got, ref[E]))
        torch.cuda.synchronize()
        out[f"splitk_{lt}"] = {"iterations": 400 * 3, "mismatches": bad, "seconds": round(time.time() - t0, 1)}
        del load
print(json.dumps(out))
