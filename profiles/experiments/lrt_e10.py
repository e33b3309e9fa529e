"""LRT BayesianAlexNet bs 512, num_ens 10: fp32 LRT path against the split-mode LRT chain, with the first layer in space-to-depth form
(ten dual contractions of the shared input) or on the fp32 kernel's shared-moments form (one dual contraction + ten samplings)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import layers  # noqa
from bbb_hip import ops, rng, ensemble as ens, zoo
import ref_port_torch as P
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "lrt", "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device="cuda")
E = 10
out = {}
with torch.no_grad():
    for name, cfg in (("fp32", dict(gemm_mode="fp32")), ("split_s2d_first", dict(gemm_mode="bf16x3")),
                      ("split_fp32_first", dict(gemm_mode="bf16x3", c8x3_s2d=False))):
        with ops.use_config(**cfg):
            pipe = ens.GraphedPipeline(net, x, E, depth=3)
            for _ in range(30):
                pipe.step()
            pipe.sync()
            t0 = time.perf_counter()
            for _ in range(90):
                pipe.step()
            pipe.sync()
            dt = (time.perf_counter() - t0) / 90
            out[name] = {"ms_per_step": round(1e3 * dt, 4), "samples_per_s": round(512 * E / dt)}
            del pipe
print(json.dumps(out))
