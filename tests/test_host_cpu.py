"""CPU-only checks of the host side: C-ABI library loads and exports every symbol of include/bbb_hip.h, the
drop-in `layers` surface matches the reference's, the product path refuses to compute without a GPU, the
noise/call-counter contract, draw sharding, and the N>1 combine over a world_size-2 gloo group."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import bbb_numpy as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-bayesiancnn_amd")
HEADER = os.path.join(ROOT, "include", "bbb_hip.h")


@pytest.fixture(scope="module")
def built():
    so = os.path.join(PKG, "bbb_hip", "libbbb_hip.so")
    if not os.path.exists(so):
        subprocess.run(["bash", os.path.join(ROOT, "build.sh")], check=True)
    return so


def test_build_script_follows_included_headers(built):
    """build.sh rebuilds an object when ANY file its last compile included is newer (hipcc -MMD dependency files), not only a
    fixed list of headers: editing csrc/pconv_body.cuh -- the dominant kernel's body -- must make pconv_gemm.o stale."""
    hdr = os.path.join(PKG, "csrc", "pconv_body.cuh")
    run = lambda: subprocess.run(["bash", os.path.join(ROOT, "build.sh")], check=True, capture_output=True, text=True,
                                 env=dict(os.environ, BBB_BUILD_DRY_RUN="1")).stdout
    before = run()
    st = os.stat(hdr)
    try:
        os.utime(hdr, None)                                  # "edited just now"
        after = run()
    finally:
        os.utime(hdr, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert "stale: build/pconv_gemm.o" in after and "pconv_gemm" not in before, (before, after)
    assert "reparam_kl" not in after                         # a source that does not include it stays as it is
    assert run() == before


def test_library_exports_every_declared_symbol(built):
    from bbb_hip import _lib
    decl = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(bbb_\w+)\s*\(", open(HEADER).read(), flags=re.M))
    assert decl, "no declarations parsed from the header"
    assert decl == set(_lib.EXPORTS), (decl ^ set(_lib.EXPORTS))
    h = _lib.lib()                      # CDLL + argtypes for every symbol; raises if one is missing
    for name in decl:
        assert hasattr(h, name)
    assert h.bbb_abi_version() == 13
    assert b"gfx950" in h.bbb_build_info()


def test_struct_layouts_match_the_header(built):
    """ctypes mirrors vs a tiny C program compiled against the header (sizeof / offsetof)."""
    from bbb_hip import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "bbb_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(bbb_segment_t), offsetof(bbb_segment_t, n), offsetof(bbb_segment_t, stream_id),
         sizeof(bbb_conv_desc_t), offsetof(bbb_conv_desc_t, x_draw_stride));
  printf("%zu %zu\n", offsetof(bbb_conv_desc_t, act), offsetof(bbb_conv_desc_t, draws));
  return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    got = [int(v) for v in out]
    S, C = _lib.Segment, _lib.ConvDesc
    assert got == [ctypes.sizeof(S), S.n.offset, S.stream_id.offset, ctypes.sizeof(C), C.x_draw_stride.offset,
                   C.act.offset, C.draws.offset]


def test_argument_errors_without_a_gpu(built):
    """Entry points validate before touching the device: error codes come back on a GPU-less host."""
    from bbb_hip import _lib
    h = _lib.lib()
    seg = (_lib.Segment * 1)()
    assert h.bbb_reparam_kl_fwd(seg, 1, 1, 0.0, 0.1, 0, 0, 0, None, None, None, None, None) == -1        # null mu
    assert h.bbb_reparam_kl_fwd(seg, 0, 1, 0.0, 0.1, 0, 0, 0, None, None, None, None, None) == -1        # nseg = 0
    assert h.bbb_reparam_kl_fwd(seg, 17, 1, 0.0, 0.1, 0, 0, 0, None, None, None, None, None) == -1       # > 16 segments
    assert h.bbb_reparam_partials(seg, 0) == -1
    d = _lib.ConvDesc()
    assert h.bbb_conv2d_fwd(ctypes.byref(d), None, None, None, None, None) == -1                   # zero geometry
    d.batch, d.cin, d.h, d.w, d.cout, d.kh, d.kw = 2, 3, 2, 2, 4, 3, 3
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = d.draws = 1
    assert h.bbb_conv2d_fwd(ctypes.byref(d), None, None, None, None, None) == -3                   # kernel > image
    d.batch = 6
    d.h = d.w = 8
    assert h.bbb_conv2d_chwn_fwd(ctypes.byref(d), None, None, None, None, None) == -3              # B % 4 != 0
    assert h.bbb_mc_tail(None, 1, 1, 1, 0, None, None) == -1
    assert h.bbb_eps_dump(None, 4, 0, 0, 0, 0, None) == -1
    assert h.bbb_maxpool_chwn(None, None, 1, 4, 4, 4, 2, 2, None) == -1


def test_product_path_refuses_cpu_tensors(built):
    import layers
    from bbb_hip import BBBHipError, ops
    layer = layers.BBB_Conv2d(3, 4, 3)
    if layer.W_mu.is_cuda:
        pytest.skip("GPU present")
    with pytest.raises(BBBHipError, match="MI355X only"):
        layer(torch.randn(2, 3, 8, 8))
    with pytest.raises(BBBHipError):
        layers.BBB_LRT_Linear(8, 4)(torch.randn(2, 8))
    with pytest.raises(BBBHipError):
        layer.kl_loss()
    with pytest.raises(BBBHipError):
        ops.mc_tail(torch.zeros(2, 3, 4))


def test_no_oracle_import_in_product():
    """The shipped package must never import / call the oracle (or the reference)."""
    bad = re.compile(r"bbb_numpy|ref_port_torch|ref_snapshot|ref_gpu_path|upstream_snapshot|_ref/|/root/reference|import\s+oracle|from\s+oracle")
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                txt = open(os.path.join(base, f)).read()
                assert not bad.search(txt), f"{f} references the oracle"


def test_launch_config_object_replaces_the_process_wide_switches():
    """ops.LaunchConfig: module attributes are views of the DEFAULT configuration; use_config is thread-local and nests."""
    import threading
    from bbb_hip import ops
    d = ops.current_config()
    assert d is ops._default_config and ops.gemm_mode == "fp32" and ops.split_k is True and ops.pool_fusion is True
    ops.gemm_mode = "bf16x3"
    try:
        assert d.gemm_mode == "bf16x3" and "gemm_mode" not in vars(ops)      # stored in the object, not as a module global
    finally:
        ops.gemm_mode = "fp32"
    with ops.use_config(gemm_mode="bf16x3", split_k=False) as c:
        assert ops.current_config() is c and c.gemm_mode == "bf16x3" and not c.split_k and d.gemm_mode == "fp32"
        assert ops.gemm_mode == "fp32"                                       # the attribute still names the default
        with ops.overlapped_launches(True) as c2:
            assert c2.launches_overlap and c2.gemm_mode == "bf16x3" and not c.launches_overlap
        seen = []
        t = threading.Thread(target=lambda: seen.append(ops.current_config()))
        t.start(); t.join()
        assert seen[0] is d                                                  # another thread: untouched
        assert ops.current_config() is c
    assert ops.current_config() is d
    assert c.key() != d.key() and d.copy().key() == d.key() and len(d.key()) == len(ops.LaunchConfig.FIELDS)
    with pytest.raises(AttributeError):
        ops.LaunchConfig(no_such_knob=1)
    with pytest.raises(AttributeError):
        ops.no_such_attribute


def test_channel_interleaved_layout_helpers_and_rule():
    """The "c8" activation layout of the bf16 path ([E][C/8][H][W][B][8], include/bbb_hip.h BBB_BF16_X_C8 / _OUT_C8): the torch
    restatement used by the tests is a permutation and its own inverse, element (e, c, h, w, b) lands at [e, c // 8, h, w, b, c % 8];
    the host rule admits exactly the layers the library has the strip form for and follows the LaunchConfig."""
    import torch
    from bbb_hip import ops
    x = torch.arange(2 * 16 * 3 * 5 * 8, dtype=torch.float32).reshape(2, 16, 3, 5, 8).to(torch.bfloat16)
    x8 = ops.to_c8(x)
    assert x8.shape == (2, 2, 3, 5, 8, 8) and x8.is_contiguous() and torch.equal(ops.from_c8(x8), x)
    assert x8[1, 1, 2, 4, 7, 5] == x[1, 13, 2, 4, 7]
    hdr = open(os.path.join(ROOT, "include", "bbb_hip.h")).read()
    assert "#define BBB_BF16_X_C8         4u" in hdr and "#define BBB_BF16_OUT_C8       8u" in hdr
    ok = ops.bf16_c8_input_ok
    assert ok((32, 5, 5), (1, 2, 1), True, False) and ok((32, 5, 5), ((1, 1), (0, 4), (1, 1)), True, False)
    assert not ok((32, 5, 5), (1, 2, 1), False, False) and not ok((32, 5, 5), (1, 2, 1), True, True)
    assert not ok((64, 5, 5), (1, 2, 1), True, False) and not ok((32, 3, 3), (1, 1, 1), True, False)
    assert not ok((32, 5, 5), (2, 2, 1), True, False) and not ok((32, 5, 5), (1, 2, 2), True, False) and not ok((32, 5, 5), (1, 5, 1), True, False)
    assert ok((32, 5, 5), (1, 2, 1), True, False, (15, 15, 256), 1)                       # default: from one step per launch on
    with ops.use_config(bf16_c8_min_items=512):
        assert not ok((32, 5, 5), (1, 2, 1), True, False, (15, 15, 256), 1)             # 15 rows x 5 strips x 2 image tiles = 150
        assert ok((32, 5, 5), (1, 2, 1), True, False, (15, 15, 256), 4)
    with ops.use_config(bf16_c8=False):
        assert not ok((32, 5, 5), (1, 2, 1), True, False)
    assert not ok((32, 5, 5), (1, 0, 1), True, False, (4, 4, 256), 16)                    # no output pixel at all


# ---------------------------------------------------------------- drop-in surface
def test_layers_surface_matches_reference_contract():
    import inspect
    import layers
    assert sorted(n for n in layers.__all__) == sorted(["BBB_Linear", "BBB_Conv2d", "BBB_LRT_Linear", "BBB_LRT_Conv2d",
                                                        "FlattenLayer", "ModuleWrapper"])
    sig = inspect.signature(layers.BBB_Conv2d.__init__)
    assert list(sig.parameters)[1:] == ["in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "bias", "priors"]
    assert [sig.parameters[k].default for k in ("stride", "padding", "dilation", "bias", "priors")] == [1, 0, 1, True, None]
    assert list(inspect.signature(layers.BBB_LRT_Conv2d.__init__).parameters)[1:] == list(sig.parameters)[1:]
    for cls in (layers.BBB_Linear, layers.BBB_LRT_Linear):
        assert list(inspect.signature(cls.__init__).parameters)[1:] == ["in_features", "out_features", "bias", "priors"]
    for cls in (layers.BBB_Conv2d, layers.BBB_Linear, layers.BBB_LRT_Conv2d, layers.BBB_LRT_Linear):
        p = inspect.signature(cls.forward).parameters
        assert list(p)[2] == "sample" and p["sample"].default is True
        assert issubclass(cls, layers.ModuleWrapper)
    conv = layers.BBB_Conv2d(3, 5, (2, 3), stride=2, padding=1, bias=False)
    assert conv.kernel_size == (2, 3) and conv.groups == 1 and conv.use_bias is False
    assert list(conv.state_dict().keys()) == ["W_mu", "W_rho"] and conv.bias_mu is None
    lin = layers.BBB_LRT_Linear(7, 3, priors={"prior_mu": 0.5, "prior_sigma": 0.2, "posterior_mu_initial": (1, 0.0),
                                              "posterior_rho_initial": (-2, 0.0)})
    assert list(lin.state_dict().keys()) == ["W_mu", "W_rho", "bias_mu", "bias_rho"]
    assert lin.prior_mu == 0.5 and lin.prior_sigma == 0.2
    assert torch.all(lin.W_mu == 1) and torch.all(lin.bias_rho == -2) and lin.W_mu.shape == (3, 7)
    np.testing.assert_allclose(lin.W_sigma.detach().numpy(), O.sigma_from_rho(lin.W_rho.detach().numpy()), rtol=1e-6)
    f = layers.FlattenLayer(12)
    assert f(torch.zeros(5, 3, 2, 2)).shape == (5, 12) and f(torch.zeros(2, 6, 2, 2)).shape == (4, 12)
    m = layers.ModuleWrapper()
    m.child = layers.ModuleWrapper()
    m.set_flag("foo", 3)
    assert m.foo == 3 and m.child.foo == 3


@pytest.mark.parametrize("name,cls_name,n_classes,cin", [("lenet", "BBBLeNet", 10, 1), ("alexnet", "BBBAlexNet", 100, 3),
                                                          ("3conv3fc", "BBB3Conv3FC", 10, 3)])
@pytest.mark.parametrize("lt", ["bbb", "lrt"])
def test_zoo_models_have_reference_state_dict(golden, name, cls_name, n_classes, cin, lt):
    """Same parameter names / shapes / init order as the upstream models: building under the seed the fixture
    was made with reproduces the reference's parameter checksums."""
    import ref_port_torch as P
    from bbb_hip import zoo
    torch.manual_seed(3)
    net = getattr(zoo, cls_name)(n_classes, cin, P.CONFIG_PRIORS, lt, "softplus")
    torch.manual_seed(3)
    params = P.init_params(name, cin, n_classes, P.CONFIG_PRIORS)
    sd = net.state_dict()
    want = [f"{n}.{k}" for n in params if not n.startswith("_") for k in ("W_mu", "W_rho", "bias_mu", "bias_rho")]
    assert list(sd.keys()) == want
    if not next(iter(sd.values())).is_cuda:
        for n in params:
            if not n.startswith("_"):
                for k in ("W_mu", "W_rho", "bias_mu", "bias_rho"):
                    assert torch.equal(sd[f"{n}.{k}"], params[n][k])
    assert net.num_classes == n_classes and net.layer_type == lt
    with pytest.raises(ValueError):
        getattr(zoo, cls_name)(10, 3, None, "nope", "softplus")
    with pytest.raises(ValueError):
        getattr(zoo, cls_name)(10, 3, None, "bbb", "tanh")


@pytest.mark.reference
def test_reference_models_build_unchanged_on_our_layers(reference_dir):
    """The UNMODIFIED upstream model files import `layers` and get ours; structure and state_dict equal the zoo's."""
    code = r'''
import sys; sys.dont_write_bytecode = True
sys.path.insert(0, "%s"); sys.path.insert(0, "%s")
import layers, torch
assert layers.__file__.startswith("%s"), layers.__file__
from models.BayesianModels.BayesianAlexNet import BBBAlexNet
from models.BayesianModels.BayesianLeNet import BBBLeNet
from models.BayesianModels.Bayesian3Conv3FC import BBB3Conv3FC
import config_bayesian as cfg
from bbb_hip import zoo
for ref, ours, cin in ((BBBAlexNet, zoo.BBBAlexNet, 3), (BBBLeNet, zoo.BBBLeNet, 1), (BBB3Conv3FC, zoo.BBB3Conv3FC, 3)):
    for lt in ("bbb", "lrt"):
        a, b = ref(10, cin, cfg.priors, lt, "softplus"), ours(10, cin, cfg.priors, lt, "softplus")
        assert [type(m).__name__ for m in a.children()] == [type(m).__name__ for m in b.children()]
        assert [n for n, _ in a.named_children()] == [n for n, _ in b.named_children()]
        assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
        assert type(a.conv1).__module__.startswith("layers.")
print("OK")
''' % (reference_dir, PKG, PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


# ---------------------------------------------------------------- rng / sharding logic
def test_call_counter_contract():
    from bbb_hip import rng
    torch.manual_seed(123)
    s, c = rng.get_state()
    assert s == 123 and c == 0
    assert rng.next_calls(3) == (123, 0) and rng.next_calls(1) == (123, 3)
    torch.manual_seed(124)                       # reseeding torch restarts the noise stream
    assert rng.get_state() == (124, 0)
    rng.manual_seed(9, call=7)
    assert rng.layer_call() == (9, 7)            # stand-alone layer: fresh call
    sc = rng.push_forward_scope()                # model forward: one call shared by all layers
    assert sc == (9, 8) and rng.layer_call() == (9, 8) and rng.layer_call() == (9, 8)
    rng.pop_forward_scope()
    assert rng.get_state() == (9, 9)


def test_stream_ids_are_per_model():
    import ref_port_torch as P
    from bbb_hip import rng, zoo
    a = zoo.BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "relu")
    b = zoo.BBBLeNet(10, 1, P.CONFIG_PRIORS, "bbb", "relu")
    assert a.conv1._stream_base != b.conv1._stream_base
    assert rng.assign_stream_ids(a) == 5 and rng.assign_stream_ids(b) == 5
    assert [m._stream_base for m in (a.conv1, a.conv2, a.fc1, a.fc2, a.fc3)] == [0, 4, 8, 12, 16]
    assert b.fc3._stream_base == 16


def test_draw_ranges_partition_the_ensemble():
    from bbb_hip.ensemble import draw_range
    for E in (1, 7, 10, 25, 80):
        for world in (1, 2, 3, 4, 8, 16):
            rs = [draw_range(E, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == E
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1


def test_group_shares_partition_the_group_of_steps():
    """ensemble.group_share: the steps * num_ens draws of a group, draw-major, in `world` contiguous ranges; what a rank is told
    about its range (first local step, local steps, offset into the first one) is consistent with the range, and the per-step
    KL shares add up to steps * num_ens forwards."""
    from bbb_hip.ensemble import group_share
    for E in (1, 3, 10):
        for G in (1, 2, 4, 5):
            for world in (1, 2, 3, 8, 64):
                seen = []
                for r in range(world):
                    lo, hi, g_lo, n_gl, off = group_share(E, G, r, world)
                    if hi <= lo:
                        assert n_gl == 0
                        continue
                    assert g_lo == lo // E and off == lo % E and 0 <= off < E
                    assert n_gl == -(-(hi - lo + off) // E) and g_lo + n_gl <= G
                    assert (hi - 1) // E == g_lo + n_gl - 1
                    seen += list(range(lo, hi))
                assert seen == list(range(G * E))


def test_pool_fusion_rule():
    """ops.pool_fusion_ok (host logic; the kernel: tests/test_gpu_pool_fusion.py): AlexNet conv1 at 40 draws fuses its MaxPool2d(2, 2),
    conv2 does not (eight 410 KB weight tiles per XCD), a lone 10-draw launch does not, a 10-draw launch beside other lanes does, a
    5-draw one does not; other pooling windows never."""
    import torch.nn as nn
    from bbb_hip import ops
    p22 = nn.MaxPool2d(2, 2)
    c1 = lambda E: ((E, 3, 32, 32, 512), (E, 64, 3, 11, 11), 4, 5, 1, E)
    assert ops.pool_fusion_ok(*c1(40), p22)
    assert not ops.pool_fusion_ok((40, 64, 4, 4, 512), (40, 192, 64, 5, 5), 1, 2, 1, 40, p22)
    assert not ops.pool_fusion_ok(*c1(10), p22)
    with ops.overlapped_launches():
        assert ops.pool_fusion_ok(*c1(10), p22) and not ops.pool_fusion_ok(*c1(5), p22)
    assert not ops.launches_overlap
    assert not ops.pool_fusion_ok(*c1(40), nn.MaxPool2d(3, 2)) and not ops.pool_fusion_ok(*c1(40), nn.MaxPool2d(2, 2, ceil_mode=True))
    assert not ops.pool_fusion_ok((40, 384, 2, 2, 512), (40, 256, 384, 3, 3), 1, 1, 1, 40, p22)      # the layer's contraction is split
    saved = ops.pool_fusion
    try:
        ops.pool_fusion = False
        assert not ops.pool_fusion_ok(*c1(40), p22)
    finally:
        ops.pool_fusion = saved


def test_conv_flops_accounting():
    from bbb_hip.ensemble import conv_flops
    # AlexNet/CIFAR bs=512 per draw: im2col 1.52 / 5.03 / 2.72 / 3.62 / 1.21 GFLOP
    shapes = [(3, 32, 64, 11, 4, 5), (64, 4, 192, 5, 1, 2), (192, 2, 384, 3, 1, 1), (384, 2, 256, 3, 1, 1), (256, 2, 128, 3, 1, 1)]
    tot_u = tot_i = 0.0
    for cin, hw, cout, k, s, p in shapes:
        u, i = conv_flops(512, cin, hw, hw, cout, k, k, s, p, 1, 1)
        tot_u += u
        tot_i += i
        assert u <= i
    assert abs(tot_i - 14.11e9) < 0.02e9
    assert abs(tot_u - 7.08e9) < 0.02e9          # more than half of the im2col matrix is padding


# ---------------------------------------------------------------- N > 1 combine over gloo (world_size 2 and 3)
_WORKER = r'''
import os, sys, math
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np, torch, torch.distributed as dist
import bbb_numpy as O
from bbb_hip import ensemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
E, B, C = int(sys.argv[3]), 6, 5
rng = np.random.default_rng(0)
logits = (rng.standard_normal((E, B, C)) * 3).astype(np.float32)          # what the E draws would produce
lo, hi = ensemble.draw_range(E, rank, world)
if hi > lo:
    ls = O.log_softmax(logits[lo:hi], axis=2).astype(np.float64)
    m = ls.max(0)
    lse = torch.tensor(m + np.log(np.exp(ls - m).sum(0)), dtype=torch.float32)   # what mc_tail(mean_over=0) returns
    kl_local = torch.tensor(123.5 * (hi - lo))
else:
    lse, kl_local = None, None
out, kl = ensemble.combine_ranks(lse, kl_local, E, dist.group.WORLD, "sum", shape=(B, C))
want = O.mc_log_outputs(logits)
assert np.allclose(out.numpy(), want, rtol=1e-5, atol=1e-6), np.abs(out.numpy() - want).max()
assert abs(kl.item() - 123.5 * E) < 1e-3
# every rank holds the same bits
g = [torch.empty_like(out) for _ in range(world)]
dist.all_gather(g, out)
assert all(torch.equal(g[0], t) for t in g)
_, klm = ensemble.combine_ranks(lse, kl_local, E, dist.group.WORLD, "mean", shape=(B, C))
assert abs(klm.item() - 123.5) < 1e-4
dist.destroy_process_group()
print("RANK_OK", rank)
'''


@pytest.mark.parametrize("world,E", [(2, 10), (2, 1), (3, 7)])
def test_ensemble_combine_over_gloo(tmp_path, world, E):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), PKG, os.path.join(ROOT, "oracle"), str(E)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, err[-3000:]


_UNIT_WORKER = r'''
import os, sys, math
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np, torch, torch.distributed as dist
import bbb_numpy as O
from bbb_hip import ensemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
E, B, C = int(sys.argv[3]), int(sys.argv[4]), 5
S = ensemble.plan_slices(E, world, B)
Bs = B // S
rng = np.random.default_rng(0)
logits = (rng.standard_normal((E, B, C)) * 3).astype(np.float32)          # what the E draws would produce for all B images
lo, hi = ensemble.unit_range(E, S, rank, world)
# what this rank's kernels produce: for every batch slice, the log-sum-exp over ITS units of that slice (-inf if none)
lse = np.full((B, C), -np.inf, dtype=np.float64)
for u in range(lo, hi):
    j, s = divmod(u, S)
    ls = O.log_softmax(logits[j, s * Bs:(s + 1) * Bs], axis=1).astype(np.float64)
    lse[s * Bs:(s + 1) * Bs] = np.logaddexp(lse[s * Bs:(s + 1) * Bs], ls)
if hi > lo:
    lse_t, kl_local = torch.tensor(lse, dtype=torch.float32), torch.tensor(123.5 * (hi - lo) / S)
else:
    lse_t, kl_local = None, None
out, kl = ensemble.combine_ranks(lse_t, kl_local, E, dist.group.WORLD, "sum", shape=(B, C))
want = O.mc_log_outputs(logits)
assert np.allclose(out.numpy(), want, rtol=1e-5, atol=2e-6), np.abs(out.numpy() - want).max()
assert abs(kl.item() - 123.5 * E) < 1e-2, kl.item()
g = [torch.empty_like(out) for _ in range(world)]
dist.all_gather(g, out)
assert all(torch.equal(g[0], t) for t in g)                                  # every rank holds the same bits
# the deal is even: no rank holds more than ceil(E*S/world) units, and the union is the whole grid
cnt = torch.tensor([hi - lo]); allc = [torch.zeros_like(cnt) for _ in range(world)]
dist.all_gather(allc, cnt)
assert sum(int(c) for c in allc) == E * S and max(int(c) for c in allc) == -(-E * S // world)
dist.destroy_process_group()
print("RANK_OK", rank, "S", S)
'''


@pytest.mark.parametrize("world,E,B,S_expected", [(2, 10, 512, 1), (3, 10, 512, 2), (4, 10, 512, 2), (8, 10, 512, 4),
                                                  (8, 25, 512, 4), (8, 3, 256, 2)])
def test_unit_sharded_ensemble_over_gloo(tmp_path, world, E, B, S_expected):
    """(draw x batch-slice) work units dealt evenly over the ranks; ONE all_gather; the combined
    log_outputs equal the single-device logmeanexp for E = 10 with half- and quarter-batch units (world 3/4 and 8)."""
    from bbb_hip import ensemble
    assert ensemble.plan_slices(E, world, B) == S_expected
    script = tmp_path / "unit_worker.py"
    script.write_text(_UNIT_WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), PKG, os.path.join(ROOT, "oracle"), str(E), str(B)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, err[-3000:]


_PROBE_WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from bbb_hip import ensemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
g = dist.group.WORLD
stalled_run = float(sys.argv[2]) > 0
stall = float(sys.argv[2]) if rank == 1 else 0.0
ensemble.capture_probe_timeout_s = 3.0
t0 = time.time()
ok = ensemble.collective_capture_ok(g, torch.device("cpu"), _force_probe=True, _stall_s=stall)
dt = time.time() - t0
assert ok is False and ensemble.last_protocol["collective"] == "eager"
assert ensemble.collective_capture_ok(g, torch.device("cpu")) is False           # cached per group OBJECT
if stalled_run:
    assert 2.5 < dt < 8.0, dt                                   # the watchdog, not the stalled rank, ended the wait
    assert "did not finish" in ensemble.last_protocol["reason"], ensemble.last_protocol
else:
    assert dt < 3.0 and "gloo" in ensemble.last_protocol["reason"], ensemble.last_protocol     # phase 1: every rank said "cannot", agreed
# the caller's own group is untouched by an abandoned probe: the eager protocol works on it
recv = torch.zeros(2 * world); send = torch.full((2,), float(rank + 1))
dist.all_gather_into_tensor(recv, send, group=g)
assert recv.tolist() == [float(r + 1) for r in range(world) for _ in range(2)]
print("RANK_OK", rank, "FALLBACK", ensemble.last_protocol["reason"], flush=True)
os._exit(0)                                                     # (an abandoned helper thread may still sit in its rendezvous)
'''


@pytest.mark.parametrize("stall", [0.0, 6.0])
def test_capture_probe_agrees_and_survives_a_stalled_rank_over_gloo(tmp_path, stall):
    """ADVICE r04 / review item 7: every rank enters the same agreement collectives whatever its local preconditions say (here:
    a host backend -> all say "cannot" and agree), and a rank that never arrives costs the others the watchdog's 3 s, after which
    they run the eager protocol on their untouched group instead of hanging."""
    script = tmp_path / "probe_worker.py"
    script.write_text(_PROBE_WORKER)
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), PKG, str(stall)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, err[-3000:]
        if stall:
            assert "FALLBACK the capture probe did not finish" in out


def test_recording_needs_evidence_that_the_nccl_event_cache_is_off():
    """the variable is read when a ProcessGroupNCCL is CONSTRUCTED; a group that predates the import (when
    the variable was not "0") still has the cache on, and re-reading the environment after our own setdefault proves nothing."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.pop("TORCH_NCCL_CUDA_EVENT_CACHE", None)
import torch, torch.distributed as dist
mode = sys.argv[1]
if mode == "group_first":
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% sys.argv[2], rank=0, world_size=1)
elif mode == "exported":
    os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] = "0"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% sys.argv[2], rank=0, world_size=1)
import bbb_hip
assert os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] == "0"
print("KNOWN_OFF", bbb_hip.nccl_event_cache_known_off())
if mode == "import_first":
    os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] = "1"
    print("AFTER_CHANGE", bbb_hip.nccl_event_cache_known_off())
''' % PKG
    import socket
    for mode, want in (("import_first", "KNOWN_OFF True"), ("group_first", "KNOWN_OFF False"), ("exported", "KNOWN_OFF True")):
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        r = subprocess.run([sys.executable, "-c", code, mode, str(port)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and want in r.stdout, (mode, r.stdout, r.stderr[-2000:])
        if mode == "import_first":
            assert "AFTER_CHANGE False" in r.stdout


def test_plan_slices_and_output_rows():
    from bbb_hip import ensemble, zoo
    assert ensemble.plan_slices(10, 1, 512) == 1
    assert ensemble.plan_slices(10, 8, 512, multiple=8) == 4
    assert ensemble.plan_slices(10, 8, 100) == 1                       # 100 images cannot be cut into aligned slices of >= 64
    # busiest rank at E=10: 8 ranks -> 5 quarter-batch units = 640 images (a perfect 1/8), 4 ranks -> 5 half-batch units
    for world, S, per in [(8, 4, 5), (4, 2, 5), (2, 1, 5)]:
        counts = [hi - lo for lo, hi in (ensemble.unit_range(10, S, r, world) for r in range(world))]
        assert max(counts) == per and sum(counts) == 10 * S
    pri = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
    net = zoo.BBBAlexNet(10, 3, pri, "bbb", "softplus")
    assert ensemble.output_rows(net, (512, 3, 32, 32)) == 512
    assert ensemble.output_rows(net, (16, 3, 224, 224)) == 16 * 49      # the view(-1, 128) quirk, layers/misc.py:35
    assert [type(m).__name__ for m in ensemble.flat_children(net)][:3] == ["BBBConv2d", "Softplus", "MaxPool2d"]


def test_cached_model_structure_follows_the_module_tree():
    """ensemble._structure: what the per-forward checks ask about a model is analysed once and cached on the module; replacing
    a direct child rebuilds it, and so does any module / parameter registration anywhere (torch's global registration hooks
    bump an epoch): nested edits and swapped Parameters are seen without invalidate()."""
    import copy
    from torch import nn
    import layers
    from bbb_hip import ensemble, zoo
    pri = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
    net = zoo.BBBLeNet(10, 1, pri, "bbb", "softplus")
    n_layers = len(ensemble.bayesian_layers(net))
    flat = ensemble.flat_children(net)
    assert ensemble.flat_children(net) is flat and n_layers == 5                    # served from the cache
    assert "_bbb_structure" not in net.state_dict() and len(net.state_dict()) == 4 * n_layers
    net.extra = layers.BBB_Linear(10, 10, priors=pri)                                # a new direct child: new signature
    assert len(ensemble.bayesian_layers(net)) == n_layers + 1 and ensemble.flat_children(net) is not flat
    twin = copy.deepcopy(net)                                                       # the copy analyses its OWN modules
    assert all(a is not b for a, b in zip(ensemble.bayesian_layers(net), ensemble.bayesian_layers(twin)))
    assert ensemble.bayesian_layers(twin)[0] is next(twin.modules().__iter__()).conv1
    seq = nn.Sequential(layers.BBB_Linear(8, 8, priors=pri), nn.ReLU())
    wrap = layers.ModuleWrapper()
    wrap.body = seq
    wrap.head = layers.BBB_Linear(8, 4, priors=pri)
    assert len(ensemble.bayesian_layers(wrap)) == 2 and len(ensemble.flat_children(wrap)) == 3
    seq.append(layers.BBB_Linear(8, 8, priors=pri))                                  # nested edit: the registration epoch moved
    assert len(ensemble.bayesian_layers(wrap)) == 3 and len(ensemble.flat_children(wrap)) == 4
    seq[2] = layers.BBB_Linear(8, 8, priors=pri)                                    # replacing a nested module
    assert ensemble.bayesian_layers(wrap)[1] is seq[2]
    old = ensemble._structure(wrap)["params"]
    wrap.head.W_mu = nn.Parameter(torch.zeros_like(wrap.head.W_mu))                 # swapping a Parameter object
    assert any(p is wrap.head.W_mu for p in ensemble._structure(wrap)["params"]) and ensemble._structure(wrap)["params"] is not old
    st = ensemble._structure(wrap)
    assert ensemble._structure(wrap) is st                                          # and nothing is rebuilt when nothing changed
    del seq[2]                                                                      # deleting goes around the hooks ...
    ensemble.invalidate(wrap)                                                       # ... so that still needs the explicit call
    assert len(ensemble.bayesian_layers(wrap)) == 2
    for p_ in wrap.parameters():
        p_.requires_grad_(False)
    assert not ensemble.any_requires_grad(wrap)                                     # flags are read live, not cached


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch, torch.distributed as dist
import ref_port_torch as P
from bbb_hip import train
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
# data-parallel step on the CPU port: every rank differentiates the ELBO of ITS shard, gradients are averaged with
# train.allreduce_gradients (several buckets forced), and must equal the single-process gradient of the full batch
torch.manual_seed(3)
params = P.init_params("lenet", 1, 10, P.CONFIG_PRIORS)
leaves = [p[k] for n, p in params.items() if not n.startswith("_") for k in ("W_mu", "W_rho", "bias_mu", "bias_rho")]
for t in leaves: t.requires_grad_(True)
Bs = 4
x = torch.rand(world * Bs, 1, 32, 32); y = torch.randint(0, 10, (world * Bs,))
def loss_of(xb, yb):
    logits, kl = P.forward("lenet", params, xb, "bbb", "softplus", sample=False)     # deterministic forward: same on all ranks
    return train.elbo(torch.log_softmax(logits, 1), yb, kl, 0.1, 1000.0)
loss_of(x, y).backward()
full = [t.grad.clone() for t in leaves]
for t in leaves: t.grad = None
loss_of(x[rank * Bs:(rank + 1) * Bs], y[rank * Bs:(rank + 1) * Bs]).backward()
n = train.allreduce_gradients(leaves, dist.group.WORLD, bucket_bytes=64 * 1024)
assert n >= 3, n                                      # LeNet's 247 KB of gradients in 64 KB buckets
for g, t in zip(full, leaves):
    assert torch.allclose(t.grad, g, rtol=2e-4, atol=1e-6 * float(g.abs().max())), float((t.grad - g).abs().max())
dist.destroy_process_group()
print("RANK_OK", rank)
'''


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_gradient_allreduce_over_gloo(tmp_path, world):
    """N1 (training extension), multi-GPU leg: shard the batch over ranks, average gradients with ONE all_reduce per
    bucket -> the gradient of the full-batch ELBO (mean-reduced NLL * train_size + beta * KL)."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), PKG, os.path.join(ROOT, "oracle")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0 and f"RANK_OK {r}" in out, err[-3000:]


@pytest.mark.reference
def test_reference_driver_imports_unchanged_through_the_launcher(reference_dir):
    """N4: run_reference.prepare() makes the UNMODIFIED main_bayesian importable on today's torch / numpy without
    torchvision, with `layers` resolving to this package; getModel builds our layers; the synthetic `data` module has the
    reference's interface.  (run() itself needs an MI355X and the checkout on the same host.)"""
    code = r'''
import sys
sys.path.insert(0, "%s")
import run_reference as rr
mb = rr.prepare("%s", synthetic=64)
import numpy as np, torch, layers
assert layers.__file__.startswith("%s")
assert np.Inf == np.inf
net = mb.getModel("alexnet", 3, 10, mb.cfg.priors, "bbb", "softplus")
assert type(net.conv1).__module__ == "layers.bbb" and type(net).__module__.startswith("models.BayesianModels")
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=6, verbose=True)          # main_bayesian.py:118 as written
import data
tr, te, cin, ncls = data.getDataset("CIFAR10")
a, b, c = data.getDataloader(tr, te, 0.2, 16, 4)
xb, yb = next(iter(a))
assert (cin, ncls) == (3, 10) and xb.shape == (16, 3, 32, 32) and len(b.dataset) == 12
assert callable(mb.train_model) and callable(mb.validate_model) and callable(mb.run)
ck = {k: tuple(v.shape) for k, v in net.state_dict().items()}
assert list(ck)[:4] == ["conv1.W_mu", "conv1.W_rho", "conv1.bias_mu", "conv1.bias_rho"]
print("OK")
''' % (PKG, reference_dir, PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


def test_metrics_helpers_match_the_oracle():
    """bbb_hip.metrics.get_beta / ELBO (host side, no device needed) against the oracle's restatement of metrics.py:12-46."""
    import bbb_numpy as O
    from bbb_hip import metrics as M
    for bt in (0.25, "Blundell", "Standard", "Soenderby", None):
        for i in range(5):
            assert M.get_beta(i, 5, bt, 3, 20) == O.get_beta(i, 5, bt, 3, 20)
    lo = torch.log_softmax(torch.randn(6, 4, generator=torch.Generator().manual_seed(0)), dim=1)
    y = torch.tensor([0, 3, 1, 1, 2, 0])
    v = M.ELBO(50)(lo, y, torch.tensor(7.0), 0.3).item()
    assert abs(v - O.elbo(lo.numpy(), y.numpy(), 7.0, 0.3, 50)) < 1e-4
    assert abs(M.acc(lo, lo.argmax(1)).item() - 1.0) < 1e-7


def test_bench_gpus_n_builds_its_own_launcher_command():
    """`python bench.py --gpus 4` with no WORLD_SIZE re-executes itself under torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1, original arguments kept (the exec itself is covered by the -m gpu contract test)."""
    import json as _json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BBB_BENCH_PRINT_LAUNCH"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    cmd = _json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"] and cmd[-7].endswith("bench.py")


def test_fast_train_static_check_rejects_pools_it_cannot_pair():
    """fast_train._train_path_static: a MaxPool2d that does not follow a Bayesian layer (+ activation) -- a
    leading pool, a pool after a pool -- and a linear layer whose in_features is not a multiple of 4 send the model to the
    reference-layout training path instead of failing inside the fused node."""
    from torch import nn
    import layers
    from bbb_hip import fast_train, zoo
    pri = {"prior_mu": 0, "prior_sigma": 0.1, "posterior_mu_initial": (0, 0.1), "posterior_rho_initial": (-5, 0.1)}
    x = torch.zeros(8, 3, 32, 32)
    assert fast_train._train_path_static(zoo.BBBAlexNet(10, 3, pri, "bbb", "softplus"), x) == "bbb"
    assert fast_train._train_path_static(zoo.BBBAlexNet(10, 3, pri, "lrt", "relu"), x) == "lrt"

    class Net(layers.ModuleWrapper):
        def __init__(self, variant):
            super().__init__()
            if variant == "leading_pool":
                self.p0 = nn.MaxPool2d(2, 2)
            self.c1 = layers.BBB_Conv2d(3, 8, 3, padding=1, priors=pri)
            self.a1 = nn.ReLU()
            self.p1 = nn.MaxPool2d(2, 2)
            if variant == "pool_pool":
                self.p2 = nn.MaxPool2d(2, 2)
            side = {"ok": 16, "leading_pool": 8, "pool_pool": 8, "odd_linear": 16}[variant]
            feat = 8 * side * side
            self.fl = layers.FlattenLayer(feat)
            if variant == "odd_linear":
                self.f0 = layers.BBB_Linear(feat, 30, priors=pri)
                self.f1 = layers.BBB_Linear(30, 10, priors=pri)
            else:
                self.f1 = layers.BBB_Linear(feat, 10, priors=pri)

    assert fast_train._train_path_static(Net("ok"), x) == "bbb"
    for bad in ("leading_pool", "pool_pool", "odd_linear"):
        assert fast_train._train_path_static(Net(bad), x) is None, bad


def test_fused_adam_state_dict_is_torch_adams():
    """The device-side learning rate of a capturable FusedAdam lives outside param_groups: state_dict() has torch.optim.Adam's
    keys only and round-trips through load_state_dict."""
    from bbb_hip import train
    p = [torch.nn.Parameter(torch.zeros(4))]
    opt = train.FusedAdam(p, lr=1e-3, capturable=True)
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3, capturable=True)
    assert set(opt.state_dict()["param_groups"][0]) <= set(ref.state_dict()["param_groups"][0])
    assert all(not torch.is_tensor(v) for v in opt.state_dict()["param_groups"][0].values())
    sd = opt.state_dict()
    sd["param_groups"][0]["lr"] = 5e-4
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 5e-4 and opt._lr_dev == {}


def test_self_capture_state_does_not_travel_with_copies_of_the_net():
    """train_step keeps its per-net capture state on the module; deepcopy / pickle of the net must not try to copy a hipGraph."""
    import copy
    import pickle
    import torch
    from bbb_hip import train
    net = torch.nn.Linear(3, 2)
    train._auto[net] = {"key": ("k",), "streak": 2, "graphed": object()}
    assert train._auto.get(net)["streak"] == 2
    twin = copy.deepcopy(net)
    assert train._auto.get(twin) is None and train._auto.get(net)["streak"] == 2
    back = pickle.loads(pickle.dumps(net))
    assert train._auto.get(back) is None
    train._auto.pop(net)
    assert train._auto.get(net) is None


def test_speculation_cache_does_not_travel_with_copies_of_the_net():
    import copy
    import pickle
    import weakref
    import torch
    import layers  # noqa: F401
    import ref_port_torch as P
    from layers import _fused
    from bbb_hip import zoo
    net = zoo.getModel("lenet", 1, 10, P.CONFIG_PRIORS, "bbb", "softplus")
    sp = net.__dict__["_bbb_spec"] = _fused._Spec()
    x = torch.zeros(2)
    sp.xref, sp.logits, sp.streak = weakref.ref(x), torch.ones(3), 5
    twin = copy.deepcopy(net)
    assert twin.__dict__["_bbb_spec"].logits is None and twin.__dict__["_bbb_spec"].streak == 0
    back = pickle.loads(pickle.dumps(net))                      # (a weak reference cannot be pickled)
    assert back.__dict__["_bbb_spec"].xref is None
    assert sp.streak == 5


def test_scratch_scope_gives_every_graph_its_own_buffers(monkeypatch):
    """ops.scratch_scope: scratch keyed by the owning graph object instead of the stream (torch's stream pool reuses
    handles; a graph replayed on another stream than it was captured on must not share tickets with a later capture)."""
    import gc
    from bbb_hip import ops
    monkeypatch.setattr(ops, "cur_stream", lambda device: 7)       # (no GPU here: one stream handle)

    class Owner:
        pass

    a, b = Owner(), Owner()
    dev = torch.device("cpu")
    with ops.scratch_scope(a):
        ka = ops._scratch_key(dev, "kl")
        with ops.scratch_scope(b):
            kb = ops._scratch_key(dev, "kl")
            tok_b = ops.current_scratch_token()
        assert ops._scratch_key(dev, "kl") == ka
    assert ka != kb and ka[2][0] == "scope" and ka[2][2] == kb[2][2] and ops.current_scratch_token() is None   # (same stream, two owners)
    with ops.scratch_scope(a):
        assert ops._scratch_key(dev, "kl") == ka               # the same owner, the same set
    with ops.scratch_scope(token=tok_b):
        assert ops._scratch_key(dev, "kl") == kb               # re-entered by token (autograd's device thread)
    ops._scratch[kb] = torch.zeros(4)
    del b
    gc.collect()
    assert kb not in ops._scratch                              # dropped with its owner


def test_outgrown_scratch_buffers_are_retired_not_freed():
    """ops._grow: a scratch buffer (split-contraction tickets / partial tiles, KL partial slots) that a later, larger launch
    outgrows stays alive -- a hipGraph captured earlier on the same stream has its address baked in."""
    from bbb_hip import ops
    key = ("test", "grow", 0)
    made = []

    def make(n):
        made.append(n)
        return torch.zeros(n, dtype=torch.uint8)

    n_ret = len(ops._retired)
    a = ops._grow(key, 100, make)
    assert ops._grow(key, 50, make) is a and ops._grow(key, 100, make) is a
    b = ops._grow(key, 101, make)
    assert b is not a and b.numel() >= 200 and ops._retired[n_ret] == (key, a) and ops._retired[n_ret][1] is a   # doubled, the old one kept
    assert ops._grow(key, 150, make) is b
    assert made == [100, 200]
    ops._scratch.pop(key)
    del ops._retired[n_ret:]
    # ... a scope's outgrown buffers go with the scope (its graphs died with the owner), and an owner that can carry neither the
    # token nor a finaliser does not get a scope of its own (it would never be released): the per-stream set instead
    class Owner:
        pass
    o = Owner()
    with ops.scratch_scope(o) as sc:
        tok = sc.tok
    sk = (0, "kl", ("scope", tok, 0))
    ops._grow(sk, 10, make)
    ops._grow(sk, 100, make)
    assert any(k == sk for k, _ in ops._retired) and sk in ops._scratch
    del o
    import gc
    gc.collect()
    assert sk not in ops._scratch and not any(k == sk for k, _ in ops._retired)
    assert ops.scratch_scope(object()).tok is None
