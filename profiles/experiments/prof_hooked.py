"""cProfile of the hooked per-layer forward (host-bound): where its ~360 us go."""
import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import layers  # noqa
from bbb_hip import zoo, rng
import ref_port_torch as P
torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device="cuda")
net.conv3.register_forward_hook(lambda m, i, o: None)
with torch.no_grad():
    for _ in range(50):
        net(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        net(x)
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
