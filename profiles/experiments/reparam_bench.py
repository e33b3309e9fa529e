"""Fused reparam+KL pass timed alone (graph of 20 launches), model size and HBM-resident sizes.
usage: BBB_HIP_LIB=<lib> python scratch/r2/reparam_bench.py <tag>"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ensemble, ops, zoo, rng

PRIORS = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = zoo.BBBAlexNet(10, 3, PRIORS, "bbb", "softplus").to(dev)
rng.assign_stream_ids(net)
mus, rhos, ids = [], [], []
for l in ensemble.bayesian_layers(net):
    m, r, i = l._param_lists()
    mus += m; rhos += r; ids += i
n_params = sum(m.numel() for m in mus)

def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e-3 / reps)
    return best

out = {"tag": sys.argv[1] if len(sys.argv) > 1 else "", "lib": os.environ.get("BBB_HIP_LIB", "default")}
with torch.no_grad():
    for E in (1, 4, 10, 25):
        t = timed(lambda: ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 1, 0, draws=E), 20)
        out[f"model_E{E}"] = {"us": round(t * 1e6, 2), "GBps": round((8 + 4 * E) * n_params / t / 1e9, 1)}
    t = timed(lambda: ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 1, 0, draws=1, sample=False, want_sigma=True, sigma_squared=True), 20)
    out["model_lrt_sigma2"] = {"us": round(t * 1e6, 2), "GBps": round(12 * n_params / t / 1e9, 1)}
    big = 1 << 26
    mu = torch.randn(big, device=dev) * 0.1
    rho = torch.randn(big, device=dev) * 0.1 - 5
    for E in (1, 4, 10):
        t = timed(lambda: ops.reparam_kl_forward([mu], [rho], 0, 0.1, [0], 1, 0, draws=E), 3)
        out[f"big_E{E}"] = {"us": round(t * 1e6, 1), "GBps": round((8 + 4 * E) * big / t / 1e9, 1)}
    dst = torch.empty_like(mu)
    t = timed(lambda: dst.copy_(mu), 3)
    out["big_copy_GBps"] = round(8 * big / t / 1e9, 1)
    # correctness spot check: KL and w deterministic, finite
    ws, _, kl = ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 7, 3, draws=2)
    ws2, _, kl2 = ops.reparam_kl_forward(mus, rhos, 0, 0.1, ids, 7, 3, draws=2)
    out["kl"] = kl.item(); out["kl_repeat_equal"] = bool(kl.item() == kl2.item())
    out["w_equal"] = all(torch.equal(a, b) for a, b in zip(ws, ws2))
    z = (ws[4] - mus[4].unsqueeze(0)) / torch.log1p(torch.exp(rhos[4])).unsqueeze(0)
    out["z_mean"] = round(z.mean().item(), 5); out["z_var"] = round(z.var().item(), 5)
print(json.dumps(out))
