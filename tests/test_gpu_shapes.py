"""The reference's own test file (tests/test_models.py upstream: shape-only smoke tests, stale at HEAD) with its intent
kept and its calls fixed: every Bayesian model x random batch size returns (logits [B, classes], kl);
FlattenLayer / conv / linear layer shapes.  Runs on the MI355X (-m gpu); the constructor-only parts also run on CPU
in tests/test_host_cpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import layers
    from bbb_hip import zoo, ensemble, rng
    return layers, zoo, ensemble, rng


@pytest.mark.parametrize("name", ["BBBLeNet", "BBBAlexNet", "BBB3Conv3FC"])
@pytest.mark.parametrize("layer_type", ["bbb", "lrt"])
def test_gpu_bayesian(mods, name, layer_type):
    layers, zoo, ensemble, rng = mods
    rs = np.random.RandomState(hash(name + layer_type) % (1 << 31))
    net = getattr(zoo, name)(10, 3, None, layer_type).cuda()           # priors=None is accepted, like upstream's layers
    for batch_size in (1, int(rs.randint(2, 256)), 255):
        batch = torch.randn((batch_size, 3, 32, 32)).cuda()
        out = net(batch)
        assert out[0].shape == (batch_size, 10)
        assert out[1].dim() == 0 and torch.isfinite(out[1]) and torch.isfinite(out[0]).all()
        # the batched ensemble path takes any batch size too (odd sizes use the reference layout)
        with torch.no_grad():
            lo, kl = ensemble.mc_forward(net, batch, 3)
        assert lo.shape == (batch_size, 10) and torch.isfinite(lo).all()
        np.testing.assert_allclose(lo.exp().sum(1).cpu().numpy(), 1.0, rtol=1e-4)


def test_flatten(mods):
    layers = mods[0]
    batch_size = np.random.randint(1, 256)
    batch = torch.randn((batch_size, 64, 4, 4)).cuda()
    out = layers.FlattenLayer(4 * 4 * 64)(batch)
    assert out.shape == (batch_size, 4 * 4 * 64)


def test_conv(mods):
    layers = mods[0]
    batch_size = np.random.randint(1, 256)
    batch = torch.randn((batch_size, 16, 24, 24)).cuda()
    for cls in (layers.BBB_Conv2d, layers.BBB_LRT_Conv2d):
        out = cls(16, 6, 4, padding=0, bias=False).cuda()(batch)
        assert out.shape == (batch_size, 6, 21, 21)


def test_linear(mods):
    layers = mods[0]
    batch_size = np.random.randint(1, 256)
    batch = torch.randn((batch_size, 128)).cuda()
    for cls in (layers.BBB_Linear, layers.BBB_LRT_Linear):
        out = cls(128, 64, bias=False).cuda()(batch)
        assert out.shape == (batch_size, 64)


def test_empty_batch_is_rejected_loudly(mods):
    layers = mods[0]
    from bbb_hip import BBBHipError
    with pytest.raises((BBBHipError, RuntimeError)):
        layers.BBB_Linear(8, 4).cuda()(torch.zeros(0, 8).cuda())
