import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops
torch.manual_seed(0)
E, Cin, H, W, B, Cout, k = 2, 64, 4, 4, 256, 192, 5
x = torch.randn(E, Cin, H, W, B, device="cuda")
w_mu = torch.randn(Cout, Cin, k, k, device="cuda") * 0.1
w_var = torch.rand(Cout, Cin, k, k, device="cuda") * 0.01
b_mu = torch.randn(Cout, device="cuda") * 0.1
b_var = torch.rand(Cout, device="cuda") * 0.01
x6 = ops.c8s3_from_f32(x, squares=True)
# input squares check
sqx = ops.c8s3_to_f32(torch.cat([x6[:, 3:], x6[:, 3:]], dim=1))
print("input squares equal:", torch.equal(sqx, x * x), float((sqx - x * x).abs().max()))
wm, wv = ops.w_tap_major(w_mu.unsqueeze(0))[0], ops.w_tap_major(w_var.unsqueeze(0))[0]
got6 = ops.lrt_conv2d_c8x3_forward(x6, wm, wv, b_mu, b_var, k, 77, 5, 6, 1, 2, 1, act="softplus")
got = ops.c8s3_to_f32(got6)
sq = ops.c8s3_to_f32(torch.cat([got6[:, 3:], got6[:, 3:]], dim=1))
d = (sq - got * got).abs()
print("output squares equal:", torch.equal(sq, got * got), float(d.max()), float((d / (got * got)).max()), int((d > 0).sum()), d.numel())
i = int(d.argmax()); print(float(got.flatten()[i]), float(sq.flatten()[i]), float((got * got).flatten()[i]))
