"""Bayes-by-Backprop layers with the local reparameterisation trick, MI355X-native.

Surface of layers/BBB_LRT/BBBConv.py:16-87 and layers/BBB_LRT/BBBLinear.py:16-79 upstream.  forward = one
parameter pass (sigma^2 + KL) + ONE dual-accumulator fp32-MFMA launch computing conv(x, mu) and
conv(x^2, sigma^2) from a single staged x tile, with act_mu + sqrt(1e-16 + act_var) * eps (eps from Philox,
indexed by output element) in its epilogue.
"""
from bbb_hip import ops, rng
from ._base import BayesianLayer


class _LRTLayer(BayesianLayer):
    def _variances(self):
        mus, rhos, _ = self._param_lists()
        if self._presampled is not None:
            out = self._presampled
            self.__dict__["_presampled"] = None
            return out
        from .misc import _refuse_low_precision
        _refuse_low_precision("a Bayesian layer called on its own")
        kl, sig2 = ops.kl_only(mus, rhos, self.prior_mu, self.prior_sigma, want_sigma=True, sigma_squared=True)
        self._take_kl(kl)
        return sig2[0], (sig2[1] if self.use_bias else None)

    def _lrt(self, x5, w_mu5, w_var5, sample, stride, padding, dilation):
        w_var, b_var = w_var5
        seed, call = rng.layer_call()
        do_sample = bool(self.training or sample)
        eps = None
        if self.eps_source is not None and do_sample:   # test / replay entry: one activation-shaped draw
            _, B, _, H, W = x5.shape
            kh, kw = w_mu5.shape[2:]
            sh, sw = ops._pair(stride)
            ph, pw = ops._pair(padding)
            dh, dw = ops._pair(dilation)
            ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
            wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
            eps = self.eps_source((B, w_mu5.shape[0], ho, wo)).to(x5.device).unsqueeze(0)
        return ops.lrt_conv2d(x5, w_mu5, w_var, self.bias_mu if self.use_bias else None, b_var, seed, call,
                              self._stream_base + 2, stride, padding, dilation, sample=do_sample, eps=eps)


class BBBConv2d(_LRTLayer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True, priors=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = (kernel_size, kernel_size)     # int only, like the reference (BBB_LRT/BBBConv.py:23)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = 1
        self._init_bayes((out_channels, in_channels, *self.kernel_size), out_channels, bias, priors)

    def forward(self, x, sample=True):
        w_var, b_var = self._variances()
        if self.eps_source is None and x.dim() == 4:
            seed, call = rng.layer_call()
            return ops.lrt_conv2d_layer(x, self.W_mu, w_var, self.bias_mu if self.use_bias else None, b_var, seed, call,
                                        self._stream_base + 2, self.stride, self.padding, self.dilation,
                                        bool(self.training or sample), None)
        y = self._lrt(x.unsqueeze(0), self.W_mu, (w_var, b_var), sample, self.stride, self.padding, self.dilation)
        return y.squeeze(0)


class BBBLinear(_LRTLayer):
    def __init__(self, in_features, out_features, bias=True, priors=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self._init_bayes((out_features, in_features), out_features, bias, priors)

    def forward(self, x, sample=True):
        w_var, b_var = self._variances()
        lead = x.shape[:-1]
        x5 = x.reshape(1, -1, self.in_features, 1, 1)
        shp = (self.out_features, self.in_features, 1, 1)
        y = self._lrt(x5, self.W_mu.reshape(shp), (w_var.reshape(shp), b_var), sample, 1, 0, 1)
        return y.reshape(*lead, self.out_features)
