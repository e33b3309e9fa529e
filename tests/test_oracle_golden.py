"""The oracle (oracle/bbb_numpy.py, oracle/ref_port_torch.py) against fixtures produced by the
unmodified reference (tests/golden/make_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

import bbb_numpy as O
import ref_port_torch as P


def _bias(L, tag, key):
    k = f"{tag}.{key}"
    return L[k] if k in L.files else None


@pytest.mark.parametrize("tag", ["bbb_conv", "bbb_conv_nb"])
def test_numpy_bbb_conv(golden, tag):
    L = golden["layers_small"]
    cin, cout, kh, kw, s, p, d, hb = L[f"{tag}.meta"]
    y, Ws, bs = O.bbb_conv2d_forward(L[f"{tag}.x"], L[f"{tag}.W_mu"], L[f"{tag}.W_rho"], _bias(L, tag, "bias_mu"),
                                     _bias(L, tag, "bias_rho"), L[f"{tag}.eps0"], _bias(L, tag, "eps1"), int(s), int(p), int(d))
    np.testing.assert_allclose(Ws, L[f"{tag}.W_sigma"], rtol=2e-7, atol=0)
    np.testing.assert_allclose(y, L[f"{tag}.y"], rtol=1e-5, atol=1e-6)
    kl = O.kl_loss(L[f"{tag}.W_mu"], Ws, 0, 0.1) + (O.kl_loss(L[f"{tag}.bias_mu"], bs, 0, 0.1) if hb else 0.0)
    assert abs(kl - float(L[f"{tag}.kl"])) <= 2e-6 * abs(kl)
    y0 = O.conv2d(L[f"{tag}.x"], L[f"{tag}.W_mu"], _bias(L, tag, "bias_mu"), int(s), int(p), int(d))
    np.testing.assert_allclose(y0, L[f"{tag}.y_nosample"], rtol=1e-5, atol=1e-6)


def test_numpy_bbb_linear(golden):
    L, tag = golden["layers_small"], "bbb_lin"
    y, Ws, bs = O.bbb_linear_forward(L[f"{tag}.x"], L[f"{tag}.W_mu"], L[f"{tag}.W_rho"], L[f"{tag}.bias_mu"],
                                     L[f"{tag}.bias_rho"], L[f"{tag}.eps0"], L[f"{tag}.eps1"])
    np.testing.assert_allclose(y, L[f"{tag}.y"], rtol=1e-5, atol=1e-6)
    kl = O.kl_loss(L[f"{tag}.W_mu"], Ws, 0, 0.1) + O.kl_loss(L[f"{tag}.bias_mu"], bs, 0, 0.1)
    assert abs(kl - float(L[f"{tag}.kl"])) <= 2e-6 * abs(kl)


def test_numpy_lrt(golden):
    L = golden["layers_small"]
    tag = "lrt_conv"
    am, av = O.lrt_moments_conv2d(L[f"{tag}.x"], L[f"{tag}.W_mu"], L[f"{tag}.W_rho"], L[f"{tag}.bias_mu"], L[f"{tag}.bias_rho"], 1, 1, 1)
    np.testing.assert_allclose(O.lrt_output(am, av, L[f"{tag}.eps0"]), L[f"{tag}.y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(am, L[f"{tag}.y_nosample"], rtol=1e-5, atol=1e-6)
    for tag in ("lrt_lin", "lrt_lin_nb"):
        am, av = O.lrt_moments_linear(L[f"{tag}.x"], L[f"{tag}.W_mu"], L[f"{tag}.W_rho"], _bias(L, tag, "bias_mu"), _bias(L, tag, "bias_rho"))
        np.testing.assert_allclose(O.lrt_output(am, av, L[f"{tag}.eps0"]), L[f"{tag}.y"], rtol=1e-5, atol=1e-6)


def test_numpy_kl_is_the_swapped_form(golden):
    Fn = golden["functions"]
    mu, sig = Fn["kl.mu"], Fn["kl.sigma"]
    np.testing.assert_allclose(O.sigma_from_rho(Fn["kl.rho"]), sig, rtol=2e-7)
    kl = O.kl_loss(mu, sig, 0, 0.1)
    assert abs(kl - float(Fn["kl.value_cfg"])) <= 2e-6 * kl
    assert abs(kl - float(Fn["kl.value_textbook"])) > 0.5 * kl      # NOT KL(q||p)
    kl3 = O.kl_loss(mu, O.sigma_from_rho(Fn["kl.rho3"]), 0, 0.1)
    assert abs(kl3 - float(Fn["kl.value_default"])) <= 2e-6 * kl3


def test_numpy_kl_grads_match_autograd(golden):
    L, tag = golden["layers_small"], "bbb_lin"
    # d/dmu, d/drho of 0.37*kl only: subtract the data term using the stored g / eps
    gmu, grho = O.kl_grads(L[f"{tag}.bias_mu"], L[f"{tag}.bias_rho"], 0, 0.1)
    g = L[f"{tag}.g"]
    gb = g.sum(0)                                     # d(sum y*g)/d bias
    sig = 1.0 / (1.0 + np.exp(-L[f"{tag}.bias_rho"].astype(np.float64)))
    np.testing.assert_allclose(gb + 0.37 * gmu, L[f"{tag}.grad_bias_mu"], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(gb * L[f"{tag}.eps1"] * sig + 0.37 * grho, L[f"{tag}.grad_bias_rho"], rtol=2e-5, atol=1e-4)


def test_numpy_tail_functions(golden):
    Fn = golden["functions"]
    np.testing.assert_allclose(O.logmeanexp(Fn["lme.x"], 2), Fn["lme.dim2"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.logmeanexp(Fn["lme.x"], 0)[None], Fn["lme.dim0_keep"], rtol=1e-6, atol=1e-6)
    v = O.elbo(Fn["elbo.log_outputs"], Fn["elbo.target"], 1234.5, 0.1, 50000)
    assert abs(v - float(Fn["elbo.value"])) <= 1e-6 * abs(v)
    for row, bt in zip(Fn["beta.table"], ["Blundell", "Soenderby", "Standard", "nonsense", 0.25]):
        assert [float(O.get_beta(b, 10, bt, 3, 40)) for b in range(4)] == list(row)


def _lenet_params(M, tag):
    names = ["conv1", "conv2", "fc1", "fc2", "fc3"]
    return {n: {k: M[f"{tag}.sd.{n}.{k}"] for k in ("W_mu", "W_rho", "bias_mu", "bias_rho")} for n in names}


def test_numpy_lenet_with_cpu_replayed_eps(golden):
    """Whole-model numpy forward with eps replayed from torch's CPU generator == reference logits."""
    M = golden["models"]
    for tag, lt, act, pri in (("lenet_bbb", "bbb", "softplus", P.CONFIG_PRIORS), ("lenet_lrt", "lrt", "relu", P.DEFAULT_PRIORS)):
        params = _lenet_params(M, tag)
        params["_prior_mu"], params["_prior_sigma"] = pri["prior_mu"], pri["prior_sigma"]
        torch.manual_seed(int(M[f"{tag}.meta"][4]) + 1)
        eps_fn = lambda name, kind, shape: torch.empty(tuple(shape)).normal_(0, 1).numpy()
        logits, kl = O.model_forward("lenet", params, M[f"{tag}.x"], lt, act, eps_fn)
        np.testing.assert_allclose(logits, M[f"{tag}.logits"], rtol=2e-4, atol=2e-5)
        assert abs(kl - float(M[f"{tag}.kl"])) <= 3e-6 * kl


CASES = [("lenet_bbb", "lenet", "bbb", "softplus", "cfg"), ("lenet_lrt", "lenet", "lrt", "relu", None),
         ("alexnet_bbb", "alexnet", "bbb", "softplus", "cfg"), ("alexnet_lrt", "alexnet", "lrt", "softplus", "cfg"),
         ("3conv3fc_bbb", "3conv3fc", "bbb", "softplus", "cfg"), ("3conv3fc_lrt", "3conv3fc", "lrt", "relu", "cfg"),
         ("alexnet224_bbb", "alexnet", "bbb", "softplus", "cfg")]


@pytest.mark.parametrize("tag,net,lt,act,pri", CASES)
def test_port_reproduces_reference_by_seed(golden, tag, net, lt, act, pri):
    """ref_port_torch draws parameters, input and eps in the reference's order: same seed -> same numbers."""
    M = golden["models"]
    ncls, cin, B, hw, seed = [int(v) for v in M[f"{tag}.meta"]]
    torch.manual_seed(seed)
    params = P.init_params(net, cin, ncls, P.CONFIG_PRIORS if pri else None)
    x = torch.rand(B, cin, hw, hw)
    cs = []
    for n in [op[1] for op in O.TOPOLOGY[net] if op[0] in ("conv", "fc")]:
        for k in ("W_mu", "W_rho", "bias_mu", "bias_rho"):
            d = params[n][k].double()
            cs.append([d.sum().item(), (d * d).sum().item()])
    np.testing.assert_allclose(np.array(cs), M[f"{tag}.checksums"], rtol=1e-12)
    torch.manual_seed(seed + 1)
    logits, kl = P.forward(net, params, x, lt, act)
    # fixtures were made single-threaded; mkldnn's accumulation order changes with the thread count
    np.testing.assert_allclose(logits.numpy(), M[f"{tag}.logits"], rtol=5e-4, atol=5e-5)
    assert abs(float(kl) - float(M[f"{tag}.kl"])) <= 1e-6 * abs(float(kl))


def test_port_mc_step(golden):
    M = golden["models"]
    torch.manual_seed(21)
    params = P.init_params("lenet", 1, 10, P.CONFIG_PRIORS)
    x = torch.rand(4, 1, 32, 32)
    labels = torch.randint(0, 10, (4,))
    torch.manual_seed(22)
    lo, kl = P.mc_step("lenet", params, x, 10, 3, "bbb", "softplus")
    np.testing.assert_allclose(lo.numpy(), M["mc_lenet.log_outputs"], rtol=1e-5, atol=1e-6)
    assert abs(float(kl) - float(M["mc_lenet.kl_sum"])) <= 1e-6 * float(kl)
    np.testing.assert_allclose(O.mc_log_outputs(M["mc_lenet.logits"]), M["mc_lenet.log_outputs"], rtol=1e-5, atol=2e-6)
    v = O.elbo(M["mc_lenet.log_outputs"], M["mc_lenet.labels"], float(M["mc_lenet.kl_sum"]) / 3, 0.1, 1000)
    assert abs(v - float(M["mc_lenet.elbo_train"])) <= 1e-6 * abs(v)


def test_philox_known_answers():
    """Random123 kat_vectors (`philox4x32 7 ...` and `philox4x32 10 ...` lines) for the round count the noise contract
    uses (7) and for Random123's default (10)."""
    def run(c, k, rounds):
        r = O.philox4x32(*[np.array([v], dtype=np.uint32) for v in c], k[0], k[1], rounds=rounds)
        return [int(v[0]) for v in r]
    pi_c, pi_k = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]
    assert O.PHILOX_ROUNDS == 7
    assert run([0, 0, 0, 0], [0, 0], 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert run([0xffffffff] * 4, [0xffffffff] * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert run(pi_c, pi_k, 7) == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    assert run([0, 0, 0, 0], [0, 0], 10) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2, 10) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run(pi_c, pi_k, 10) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_box_muller_definition():
    """The uniform construction of the noise contract: 23-bit mantissas, u1 in (0, 1], u2 in [0, 1)."""
    xa = np.array([0, 0xFFFFFFFF, 0x80000000, 0x000001FF], dtype=np.uint32)
    xb = np.array([0, 0, 0x40000000, 0xFFFFFFFF], dtype=np.uint32)
    z0, z1 = O.box_muller(xa, xb)
    assert z0[0] == 0 and z1[0] == 0                                  # u1 = 1 -> radius 0
    np.testing.assert_allclose(z0[1], np.sqrt(-2 * np.log(2.0 ** -23)), rtol=1e-6)   # smallest u1, angle 0: 5.65 sigma
    np.testing.assert_allclose(z1[2], np.sqrt(-2 * np.log(0.5)), rtol=1e-6)          # u1 = 1/2, quarter turn
    assert abs(z0[2]) < 1e-6
    assert z0[3] == 0 and z1[3] == 0                                  # the low 9 bits of a word are not used


def test_eps_stream_moments_and_windows():
    e = O.normal_eps(1234, 5, 3, 400000)
    assert abs(e.mean()) < 0.006 and abs(e.std() - 1) < 0.005
    assert abs(np.mean(e ** 3)) < 0.02 and abs(np.mean(e ** 4) - 3) < 0.05
    np.testing.assert_array_equal(O.normal_eps(1234, 5, 3, 11, start=6), e[6:17])
    assert not np.array_equal(O.normal_eps(1234, 6, 3, 16), e[:16])
    assert not np.array_equal(O.normal_eps(1234, 5, 4, 16), e[:16])


@pytest.mark.reference
def test_port_matches_live_reference(reference_dir):
    """The port against the live, unmodified upstream modules."""
    import subprocess
    code = r'''
import sys; sys.dont_write_bytecode=True
sys.path.insert(0, "%s"); sys.path.insert(0, "%s")
import torch, numpy as np
import ref_port_torch as P
from models.BayesianModels.BayesianAlexNet import BBBAlexNet
import config_bayesian as cfg
torch.manual_seed(5); net = BBBAlexNet(10, 3, cfg.priors, "bbb", "softplus"); x = torch.rand(4,3,32,32)
torch.manual_seed(6); a, ka = net(x)
torch.manual_seed(5); params = P.init_params("alexnet", 3, 10, P.CONFIG_PRIORS); x2 = torch.rand(4,3,32,32)
torch.manual_seed(6); b, kb = P.forward("alexnet", params, x2, "bbb", "softplus")
assert torch.equal(a, b) and torch.equal(ka, kb), (float((a-b).abs().max()), float(ka), float(kb))
print("OK")
''' % (reference_dir, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_numpy_uncertainty_vs_reference(golden):
    """oracle.uncertainty == uncertainty_estimation.get_uncertainty_per_image of the reference (softmax and normalized)."""
    U = golden["uncertainty"]
    for tag in ("lrt", "bbb"):
        for norm in (0, 1):
            k = f"unc_{tag}_{norm}"
            logits = U[k + ".logits"]                       # [T, C]: T rows of one batch = T "draws" of one image
            pred, epi, ale = O.uncertainty(logits[:, None, :], normalized=bool(norm))
            np.testing.assert_allclose(pred[0], U[k + ".pred"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(epi[0], U[k + ".epistemic"], rtol=2e-4, atol=1e-9)
            np.testing.assert_allclose(ale[0], U[k + ".aleatoric"], rtol=2e-5, atol=1e-8)
    # with BBB layers all T rows share one weight draw: epistemic is (numerically) zero -- SURVEY.md section 8f N2
    assert U["unc_bbb_0.epistemic"].max() < 1e-10 and U["unc_lrt_0.epistemic"].max() > 1e-8


# ---------------------------------------------------------------- bf16 storage model (no upstream counterpart)
def test_bf16_storage_model_is_a_small_perturbation_of_the_fp32_oracle():
    """oracle.model_forward_bf16 = the pinned fp32 oracle with documented rounding points: rounding is torch's
    nearest-even, and the logits stay within 2e-2 of the fp32 oracle's scale under the same noise."""
    import torch
    rs = np.random.RandomState(0)
    a = (rs.randn(20000) * 10 ** rs.uniform(-6, 6, 20000)).astype(np.float32)
    assert np.array_equal(O.bf16_round(a), torch.from_numpy(a).to(torch.bfloat16).float().numpy())
    assert np.array_equal(O.bf16_round(O.bf16_round(a)), O.bf16_round(a))
    torch.manual_seed(5)
    params = P.init_params("lenet", 1, 10, P.CONFIG_PRIORS)
    npar = {n: {k: v.numpy() for k, v in p.items()} for n, p in params.items() if not n.startswith("_")}
    npar["_prior_mu"], npar["_prior_sigma"] = params["_prior_mu"], params["_prior_sigma"]
    x = rs.rand(4, 1, 32, 32).astype(np.float32)
    names = [op[1] for op in O.TOPOLOGY["lenet"] if op[0] in ("conv", "fc")]

    def eps_fn(name, kind, shape):
        return O.normal_eps(77, 3, 4 * names.index(name) + (0 if kind == "W" else 1), int(np.prod(shape))).reshape(shape)
    want, kl = O.model_forward("lenet", npar, x, "bbb", "softplus", eps_fn)
    got, kl16 = O.model_forward_bf16("lenet", npar, x, "softplus", eps_fn)
    assert kl16 == kl
    assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max()
    assert np.abs(got - want).max() > 0


@pytest.mark.reference
def test_port_times_like_the_live_reference(reference_dir):
    """bench.py's cpu_baseline uses the port where /root/reference does not exist (the GPU box): besides producing the same
    bits, the port must COST the same as the unmodified modules on the same cores.  One AlexNet draw at bs=256, interleaved
    runs, minimum of 9 each, within 5 % in at least one of up to eight attempts (shared build hosts are noisy: a neighbour's
    burst shifts one attempt by 10 % or more, a real difference between the two code paths shifts all of them)."""
    import subprocess
    code = r"""
import sys, time; sys.dont_write_bytecode = True
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import ref_port_torch as P
from models.BayesianModels.BayesianAlexNet import BBBAlexNet
torch.manual_seed(0)
net = BBBAlexNet(10, 3, P.CONFIG_PRIORS, "bbb", "softplus")
params = P.init_params("alexnet", 3, 10, P.CONFIG_PRIORS)
x = torch.rand(256, 3, 32, 32)
def t_ref():
    t = time.perf_counter(); net(x); return time.perf_counter() - t
def t_port():
    t = time.perf_counter(); P.forward("alexnet", params, x, "bbb", "softplus"); return time.perf_counter() - t
with torch.no_grad():
    for _ in range(2): t_ref(); t_port()
    best = None
    for attempt in range(8):
        a, b = [], []
        for i in range(9):
            if i %% 2: a.append(t_ref()); b.append(t_port())
            else: b.append(t_port()); a.append(t_ref())
        r = min(b) / min(a)
        best = r if best is None or abs(r - 1) < abs(best - 1) else best
        if abs(r - 1) <= 0.05: break
print("RATIO %%.4f" %% best)
""" % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), reference_dir)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ratio = float(r.stdout.split("RATIO")[1])
    assert abs(ratio - 1.0) <= 0.05, f"port / reference time ratio {ratio}"
