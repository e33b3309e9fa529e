"""TEST INFRASTRUCTURE (oracle side; executed by bench.py's baseline leg as a subprocess, never imported by the product).

The UNMODIFIED upstream modules on the same MI355X through PyTorch-ROCm's own kernels (ATen / MIOpen / rocBLAS; the layers'
eps still comes from the CPU generator and is copied over, layers/BBB/BBBConv.py:63) -- the reference's GPU path, as a user of the
reference gets it on this box -- next to this project's step.  Metric step: BayesianAlexNet, bs 512, num_ens 10, bbb, forward only
(main_bayesian.py:73-80).  Test infrastructure: reads oracle/_ref (never part of the product)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_snapshot
BUDGET_S = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
kind, ref = ref_snapshot.checkout()
assert ref is not None, "no upstream files"
sys.dont_write_bytecode = True
sys.path.insert(0, ref)
import torch
import torch.nn.functional as F
from models.BayesianModels.BayesianAlexNet import BBBAlexNet
import utils as ref_utils
import layers
assert layers.__file__.startswith(ref)
B, C, E = 512, 10, 10
pri = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
torch.manual_seed(0)
t0 = time.perf_counter()
net = BBBAlexNet(C, 3, pri, "bbb", "softplus").to("cuda:0")
x = torch.rand(B, 3, 32, 32, device="cuda:0")
out = {"upstream": kind}


def step():
    outputs = torch.zeros(B, C, E, device="cuda:0")
    kl = 0.0
    for j in range(E):
        net_out, _kl = net(x)
        kl += _kl
        outputs[:, :, j] = F.log_softmax(net_out, dim=1).data
    return ref_utils.logmeanexp(outputs, dim=2), kl


for mode, ctx in (("no_grad", torch.no_grad), ("autograd_enabled", torch.enable_grad)):
    with ctx():
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        first = time.perf_counter() - t1                 # includes MIOpen's kernel selection / compilation for the five conv shapes
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        n, t2 = 0, time.perf_counter()
        while time.perf_counter() - t2 < BUDGET_S:
            step()
            n += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t2) / n
    out[mode] = {"first_step_s": round(first, 2), "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B * E / dt, 1), "steps_timed": n}
print(json.dumps(out))
