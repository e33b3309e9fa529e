"""Bayes-by-Backprop layers with weight-space reparameterisation, MI355X-native.

Same constructor / forward(x, sample=True) / kl_loss() / state_dict surface as the reference's
layers/BBB/BBBConv.py:14-83 and layers/BBB/BBBLinear.py:14-76.  forward = one fused reparam+KL launch
(eps, softplus, mu + sigma*eps, KL in a single pass over (mu, rho)) + one fp32-MFMA implicit-GEMM launch.
"""
from bbb_hip import ops, rng
from ._base import BayesianLayer


class _BBBLayer(BayesianLayer):
    def _weights(self, sample):
        """-> (w [1, ...], bias [1, Cout] | None), caching this call's KL."""
        mus, rhos, ids = self._param_lists()
        if self._presampled is not None:
            w, b = self._presampled
            self.__dict__["_presampled"] = None
            return w, b
        from .misc import _refuse_low_precision
        _refuse_low_precision("a Bayesian layer called on its own")
        if self.training or sample:
            seed, call = rng.layer_call()
            eps = None
            if self.eps_source is not None:      # test / replay entry: external noise, W first then bias
                eps = [self.eps_source(tuple(m.shape)).to(m.device).unsqueeze(0) for m in mus]
            kl, ws = ops.sample_weights(mus, rhos, self.prior_mu, self.prior_sigma, ids, seed, call, 1, eps=eps)
            self._take_kl(kl)
            return ws[0], (ws[1] if self.use_bias else None)
        kl, _ = ops.kl_only(mus, rhos, self.prior_mu, self.prior_sigma)
        self._take_kl(kl)
        return self.W_mu.unsqueeze(0), (self.bias_mu.unsqueeze(0) if self.use_bias else None)


class BBBConv2d(_BBBLayer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True, priors=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = 1
        self._init_bayes((out_channels, in_channels, *self.kernel_size), out_channels, bias, priors)

    def forward(self, input, sample=True):
        w, b = self._weights(sample)
        return ops.conv2d_layer(input, w, b, self.stride, self.padding, self.dilation)


class BBBLinear(_BBBLayer):
    def __init__(self, in_features, out_features, bias=True, priors=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self._init_bayes((out_features, in_features), out_features, bias, priors)

    def forward(self, input, sample=True):
        w, b = self._weights(sample)
        lead = input.shape[:-1]
        x = input.reshape(1, -1, self.in_features, 1, 1)
        y = ops.conv2d(x, w.reshape(1, self.out_features, self.in_features, 1, 1), b, 1, 0, 1)
        return y.reshape(*lead, self.out_features)
