"""How many workgroups of a GEMM launch are resident per CU over time (needs a -DPCONV_STAMPS build via BBB_HIP_LIB): every
workgroup records start / end (100 MHz wall clock) and its hardware ids."""
import ctypes, os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "pytorch-bayesiancnn_amd"))
import torch
from bbb_hip import ops
lib = ctypes.CDLL(os.environ["BBB_HIP_LIB"])
lib.bbb_census_set.argtypes = [ctypes.c_void_p]
B = 512
L = [("conv1", 3, 32, 32, 64, 11, 4, 5), ("conv2", 64, 4, 4, 192, 5, 1, 2), ("conv4", 384, 2, 2, 256, 3, 1, 1), ("conv5", 256, 2, 2, 128, 3, 1, 1)]
ops.split_k = False
buf = torch.zeros(1 << 18, dtype=torch.int64, device="cuda")
for E in (10, 40):
    for name, Cin, H, W, Cout, k, st, pd in L:
        x = torch.randn(1 if name == "conv1" else E, Cin, H, W, B, device='cuda'); w = torch.randn(E, Cout, Cin, k, k, device='cuda') * 0.05; b = torch.zeros(E, Cout, device='cuda')
        for _ in range(2): ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
        torch.cuda.synchronize()
        buf.zero_(); lib.bbb_census_set(ctypes.c_void_p(buf.data_ptr()))
        ops.conv2d_chwn_forward(x, w, b, st, pd, 1, act="softplus")
        torch.cuda.synchronize()
        lib.bbb_census_set(ctypes.c_void_p(0))
        r = buf.view(-1, 4).cpu()
        r = r[r[:, 1] > 0]
        t0 = int(r[:, 0].min()); t1 = int(r[:, 1].max())
        span = (t1 - t0) * 0.01
        life = ((r[:, 1] - r[:, 0]).double() * 0.01)
        key = (r[:, 3] & 7) * 4096 + ((r[:, 2] >> 13) & 7) * 64 + ((r[:, 2] >> 8) & 15)
        cus = len(set(key.tolist()))
        cyc = (r[:, 3] >> 8).double()
        ghz = (cyc / ((r[:, 1] - r[:, 0]).double() * 10.0))          # cycle-counter ticks per ns of wall clock, per workgroup
        # concurrency over time in 5 us buckets
        nb = int(span // 5) + 1
        conc = [0.0] * nb
        for s_, e_ in zip(((r[:, 0] - t0).double() * 0.01).tolist(), ((r[:, 1] - t0).double() * 0.01).tolist()):
            k0 = int(s_ // 5)
            while k0 * 5 < e_ and k0 < nb:
                conc[k0] += (min(e_, (k0 + 1) * 5) - max(s_, k0 * 5)) / 5
                k0 += 1
        per_cu = {}
        for kk in key.tolist(): per_cu[kk] = per_cu.get(kk, 0) + 1
        print(json.dumps({"E": E, "layer": name, "wgs": int(r.shape[0]), "span_us": round(span, 1), "cus_seen": cus,
                          "mean_resident_per_cu": round(life.sum().item() / span / max(cus, 1), 2),
                          "wg_life_us_mean_min_max": [round(life.mean().item(), 1), round(life.min().item(), 1), round(life.max().item(), 1)],
                          "wgs_per_cu_min_max": [min(per_cu.values()), max(per_cu.values())],
                          "cycle_counter_GHz_mean_min_max": [round(ghz.mean().item(), 3), round(ghz.min().item(), 3), round(ghz.max().item(), 3)],
                          "resident_total_over_time": [round(v) for v in conc[:12]]}), flush=True)
