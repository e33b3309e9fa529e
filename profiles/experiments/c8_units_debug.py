"""Where does a work-unit launch of the c8 chain leave the whole step's bits?  Records every launch's output in the whole step and
in a rank's share and compares unit by unit.  (Debug aid of round 6.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import layers  # noqa: F401,E402
from bbb_hip import ops, rng, ensemble as ens, zoo  # noqa: E402
import ref_port_torch as P  # noqa: E402

torch.manual_seed(0)
net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").cuda()
rng.assign_stream_ids(net)
x = torch.rand(512, 3, 32, 32, device="cuda")
E, S = 10, 2

names = ["conv2d_c8x3_forward", "conv2d_chwn_forward", "maxpool_c8s3", "c8s3_from_f32", "sample_weights_tm"]
log = []
orig = {n: getattr(ops, n) for n in names}


def wrap(n):
    def f(*a, **k):
        y = orig[n](*a, **k)
        log.append((n, y, {kk: vv for kk, vv in k.items() if kk in ("units", "n_units", "pool", "bf16x3", "tile")}))
        return y
    return f


for n in names:
    setattr(ops, n, wrap(n))


def f32(n, t):
    if n in ("maxpool_c8s3", "c8s3_from_f32") or (n == "conv2d_c8x3_forward" and t.dtype == torch.bfloat16):
        return orig_to(t)
    return t


orig_to = ops.c8s3_to_f32

with torch.no_grad(), ops.use_config(gemm_mode="bf16x3", s3_min_images=0):
    full, kl = ens._mc_logits_chwn(net, x, E, 7, 3)
    full_log, log = log, []
    lo, hi = ens.unit_range(E, S, 0, 4)
    part, klp = ens._mc_logits_chwn(net, x, E, 7, 3, units=(S, lo, hi))
    part_log = log
    print("units", lo, hi, "launches", len(full_log), len(part_log))
    for (n, yf, kf), (n2, yp, kp) in zip(full_log, part_log):
        assert n == n2, (n, n2)
        if n == "sample_weights_tm":
            klf, wf = yf
            klq, wq = yp
            print(n, "kl equal", torch.equal(klf, klq), [torch.equal(a[:b.shape[0]], b) for a, b in zip(wf, wq)])
            continue
        a, b = f32(n, yf), f32(n, yp)           # [E, C, H, W, B] / [U, C, H, W, B/S]
        bad = 0
        for i, u in enumerate(range(lo, hi)):
            j, sl = divmod(u, S)
            ref = a[j][..., sl * 256:(sl + 1) * 256]
            if not torch.equal(b[i], ref):
                bad += 1
        print(n, tuple(a.shape), tuple(b.shape), kf, kp, "units differing:", bad)
    # the tile claim: 128 / 256 images per workgroup, same bits
    xx = ops.c8s3_from_f32(torch.randn(4, 64, 4, 4, 512, device="cuda"))
    w = torch.randn(4, 192, 25, 64, device="cuda") * 0.05
    y1 = orig["conv2d_c8x3_forward"](xx, w, None, 5, 1, 2, 1, tile=128)
    y2 = orig["conv2d_c8x3_forward"](xx, w, None, 5, 1, 2, 1, tile=256)
    print("tile 128 == 256:", torch.equal(y1, y2))
