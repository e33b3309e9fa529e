"""Seeded random-geometry sweep of the three implicit-GEMM kernels (NCHW fp32, batch-innermost fp32, batch-innermost
bf16) against the fp64 oracle convolution: ragged channel / batch tiles, strides, dilations, asymmetric kernels and
paddings larger than the kernel reach.  Tolerances as in test_gpu_kernels.py / test_gpu_bf16.py.  Run with -m gpu."""
import numpy as np
import pytest
import torch

import bbb_numpy as O

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        kh, kw = int(rs.choice([1, 2, 3, 5, 7])), int(rs.choice([1, 2, 3, 5]))
        sh, sw = int(rs.choice([1, 1, 2, 3])), int(rs.choice([1, 1, 2]))
        dh, dw = int(rs.choice([1, 1, 2])), int(rs.choice([1, 1, 2]))
        ph, pw = int(rs.randint(0, 4)), int(rs.randint(0, 4))
        H, W = int(rs.randint(1, 13)), int(rs.randint(1, 13))
        if H + 2 * ph < dh * (kh - 1) + 1 or W + 2 * pw < dw * (kw - 1) + 1:
            continue
        Cin, Cout = int(rs.choice([1, 3, 8, 17, 64])), int(rs.choice([1, 10, 64, 65, 130]))
        B = int(rs.choice([8, 16, 40, 136, 264]))
        E = int(rs.choice([1, 2, 3]))
        out.append((B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, E, bool(rs.randint(0, 2)), str(rs.choice(["none", "relu", "softplus"]))))
    return out


CASES = _cases(24, 20260923)


def _ref(x, w, b, c, e, xs, act):
    B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, E, _, _ = c
    xe = x[0 if xs else e]
    pre = O.conv2d(xe, w[e], b[e], (sh, sw), (ph, pw), (dh, dw))
    mag = O.conv2d(np.abs(xe), np.abs(w[e]), np.abs(b[e]), (sh, sw), (ph, pw), (dh, dw))
    y = {"none": lambda v: v, "relu": O.relu_act, "softplus": O.softplus_act}[act](pre)
    return y, mag


@pytest.mark.parametrize("c", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_random_geometry_all_kernels(c):
    from bbb_hip import ops
    B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, E, xs, act = c
    torch.manual_seed(1000 + CASES.index(c))
    x = torch.randn(1 if xs else E, B, Cin, H, W, device="cuda")
    w = torch.randn(E, Cout, Cin, kh, kw, device="cuda") * 0.3
    b = torch.randn(E, Cout, device="cuda")
    a = None if act == "none" else act
    geom = ((sh, sw), (ph, pw), (dh, dw))
    y_nchw = ops.conv2d_forward(x, w, b, *geom, act=a)                                        # [E, B, Cout, Ho, Wo]
    xc = x.permute(0, 2, 3, 4, 1).contiguous()
    y_chwn = ops.conv2d_chwn_forward(xc, w, b, *geom, act=a).permute(0, 4, 1, 2, 3)
    xn, wn, bn = x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy()
    for e in range(E):
        want, mag = _ref(xn, wn, bn, c, e, xs, act)
        tol = 2e-5 * mag + 2e-6
        for name, got in (("nchw", y_nchw[e]), ("chwn", y_chwn[e])):
            g = got.cpu().numpy()
            assert g.shape == want.shape, (name, g.shape, want.shape)
            err = np.abs(g - want)
            assert (err <= tol).all(), f"{name} draw {e}: excess {(err - tol).max():.3e}"
    # bf16 storage variant: same bf16 operands on both sides, bf16 output
    K = Cin * kh * kw
    Kp = (K + 7) & ~7
    wb = torch.zeros(E, Cout, Kp, dtype=torch.bfloat16, device="cuda")
    wb[:, :, :K] = w.reshape(E, Cout, K).to(torch.bfloat16)
    xb = xc.to(torch.bfloat16)
    outs = [("bf16", ops.conv2d_chwn_bf16_forward(xb, wb, b, (Cin, kh, kw), *geom, act=a))]
    if Cin % 8 == 0:                               # tap-major rows: padding taps skipped instead of multiplied by zeros
        wt = torch.zeros_like(wb)
        wt[:, :, :K] = w.permute(0, 1, 3, 4, 2).reshape(E, Cout, K).to(torch.bfloat16)
        outs.append(("bf16-tap-major", ops.conv2d_chwn_bf16_forward(xb, wt, b, (Cin, kh, kw), *geom, act=a, tap_major=True)))
    xr = xb.float().permute(0, 4, 1, 2, 3).cpu().numpy()
    wr = w.to(torch.bfloat16).float().cpu().numpy()
    for name, y16 in outs:
        y16 = y16.float().permute(0, 4, 1, 2, 3)
        for e in range(E):
            want, mag = _ref(xr, wr, bn, c, e, xs, act)
            tol = 2e-5 * mag + 2e-6 + np.abs(want) * 2.0 ** -8
            err = np.abs(y16[e].cpu().numpy() - want)
            assert (err <= tol).all(), f"{name} draw {e}: excess {(err - tol).max():.3e}"
