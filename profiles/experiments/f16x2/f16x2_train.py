"""Training step (AlexNet bs 512, num_ens 10, forward + backward + Adam) with ops.gemm_mode = "fp16x2".  First version of the mode
(every batch-innermost GEMM launch, including the backward's role-swapped ones): 3.10 -> 2.44 ms per step, but gradients up to 41 %
off -- the gradient operands (1e-6-sized) lie far below the split's operand window.  The mode is therefore restricted to the
inference ensemble path; this script now shows identical gradients and step times in both modes.
    python profiles/experiments/f16x2_train.py [eager|auto] [modes...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
import torch.nn.functional as F
import bench
from bbb_hip import ensemble, ops, rng, train
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = bench.CONFIGS["metric"]
net, x = bench.build_net(cfg, dev)
y = torch.randint(0, 10, (512,), device=dev)
GRAPH = None if (len(sys.argv) > 1 and sys.argv[1] == "auto") else False
MODES = sys.argv[2:] or ["fp32", "fp16x2"]
grads = {}
for mode in ("fp32", "fp16x2"):
    ops.gemm_mode = mode
    net.zero_grad(set_to_none=True)
    rng.manual_seed(5, call=0)
    lo, kl = ensemble.mc_forward(net, x, 10, kl_mode="mean")
    (F.nll_loss(lo, y) * 50000.0 + 0.1 * kl).backward()
    grads[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
worst = max(float((grads["fp16x2"][n] - grads["fp32"][n]).abs().max() / (grads["fp32"][n].abs().max() + 1e-20)) for n in grads["fp32"])
row = {"max_rel_grad_diff": worst}
for mode in MODES:
    ops.gemm_mode = mode
    opt = train.FusedAdam(net.parameters(), lr=1e-4)
    for _ in range(6):
        train.train_step(net, opt, x, y, 10, 0.1, 50000.0, graph=GRAPH)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        train.train_step(net, opt, x, y, 10, 0.1, 50000.0, graph=GRAPH)
    torch.cuda.synchronize()
    row[mode + "_ms_per_step"] = round((time.perf_counter() - t0) / 30 * 1e3, 3)
ops.gemm_mode = "fp32"
print(json.dumps(row))
