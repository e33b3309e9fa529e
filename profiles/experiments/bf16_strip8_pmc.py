"""A few eager launches of 3Conv3FC conv2 (bf16, bs 256, 16 steps per launch) on the general kernel and on the strip form over
channel-interleaved input, for a rocprofv3 --pmc pass (matrix-pipe / VALU busy cycles per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops
dev = torch.device("cuda:0")
x = torch.rand(16, 32, 15, 15, 256, device=dev).to(torch.bfloat16)
x8 = ops.to_c8(x)
w = (torch.randn(16, 64, 800, device=dev) * 0.03).to(torch.bfloat16)
b = torch.randn(16, 64, device=dev) * 0.1
with torch.no_grad():
    for _ in range(6):
        ops.conv2d_chwn_bf16_forward(x, w, b, (32, 5, 5), 1, 2, 1, act="softplus", tap_major=True)
        ops.conv2d_chwn_bf16_forward(x8, w, b, (32, 5, 5), 1, 2, 1, act="softplus", tap_major=True)
torch.cuda.synchronize()
