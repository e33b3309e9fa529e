"""Host cost of a launch through the ctypes binding (review r05 item 10: "measure first"): us per call of
  (a) a no-argument C entry (the ctypes floor), (b) bbb_conv2d_chwn_fwd called directly with prebuilt arguments (marshalling of a
  descriptor pointer, five pointers and a stream), (c) ops.conv2d_chwn_forward (the Python wrapper: checks, descriptor, output
  allocation, the call), (d) a torch op on the same tensor (ATen's own dispatch, for scale), (e) one eager per-layer forward of
  BayesianAlexNet with a forward hook on conv3 (the "hooked" path of bench.py's slow_paths) and its launches.
A torch-extension shim over the same C ABI would replace (b)'s marshalling only -- (c) minus (b) is Python either way."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-bayesiancnn_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import layers  # noqa: F401
from bbb_hip import ops, _lib, zoo, rng

dev = torch.device("cuda:0"); torch.cuda.set_device(0)
L = _lib.lib()
N = 4000


def per_call(fn, n=N, sync_every=500):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    done = 0
    while done < n:
        k = min(sync_every, n - done)
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        t += time.perf_counter() - t0
        torch.cuda.synchronize()
        done += k
    return round(1e6 * t / n, 2)


out = {}
out["abi_version_call_us"] = per_call(L.bbb_abi_version)
with torch.no_grad():
    x = torch.rand(1, 8, 4, 4, 8, device=dev)
    w = torch.rand(1, 8, 8, 3, 3, device=dev)
    b = torch.rand(1, 8, device=dev)
    y = ops.conv2d_chwn_forward(x, w, b, 1, 1, 1)
    d, ho, wo = ops._desc_chwn(x, w, 1, 1, 1, 1, False, False, None)
    st = ops.cur_stream(dev)
    args = (ctypes.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st)
    out["direct_ctypes_conv_call_us"] = per_call(lambda: L.bbb_conv2d_chwn_fwd(*args))
    out["ops_wrapper_conv_call_us"] = per_call(lambda: ops.conv2d_chwn_forward(x, w, b, 1, 1, 1))
    out["ops_wrapper_conv_call_out_us"] = per_call(lambda: ops.conv2d_chwn_forward(x, w, b, 1, 1, 1, out=y))
    t = torch.rand(1024, device=dev)
    out["torch_relu_call_us"] = per_call(lambda: torch.relu_(t))
    out["torch_empty_call_us"] = per_call(lambda: torch.empty((1, 8, 4, 4, 8), device=dev), sync_every=4000)

    import ref_port_torch as P
    torch.manual_seed(0)
    net = zoo.getModel("alexnet", 3, 10, P.CONFIG_PRIORS, "bbb", "softplus").to(dev)
    rng.assign_stream_ids(net)
    xb = torch.rand(512, 3, 32, 32, device=dev)
    seen = []
    h = net.conv3.register_forward_hook(lambda m, i, o: seen.append(1))
    launches = [0]
    names = [n for n in dir(L) if n.startswith("bbb_")]
    out["hooked_forward_us"] = per_call(lambda: net(xb), n=200, sync_every=50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        net(xb)
    torch.cuda.synchronize()
    out["hooked_forward_wall_us"] = round(1e6 * (time.perf_counter() - t0) / 50, 1)
    h.remove()
print(json.dumps(out))
