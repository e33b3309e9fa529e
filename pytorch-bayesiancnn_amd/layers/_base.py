"""Shared plumbing of the four Bayesian layers: priors, parameters, noise-stream ids, KL cache."""
import torch
from torch.nn import Parameter

from bbb_hip import ops, rng
from .misc import ModuleWrapper

_DEFAULT_PRIORS = {
    "prior_mu": 0,
    "prior_sigma": 0.1,
    "posterior_mu_initial": (0, 0.1),
    "posterior_rho_initial": (-3, 0.1),
}


def default_device():
    # the reference pins cuda:0 (layers/BBB/BBBConv.py:27); parameters may later be moved with .to()
    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


class BayesianLayer(ModuleWrapper):
    """State common to BBB and BBB_LRT layers.

    Parameters W_mu, W_rho, bias_mu, bias_rho (bias pair registered as None without bias) -> the same
    state_dict keys as the reference.  Departures, both deliberate: eps comes from the on-chip Philox
    stream (rng.py) instead of torch's CPU generator, and every tensor follows the parameters' device
    rather than a hard-coded cuda:0 (needed for one-process-per-GPU jobs).
    """

    def _init_bayes(self, weight_shape, n_out, bias, priors):
        self.use_bias = bias
        self.device = default_device()
        priors = dict(_DEFAULT_PRIORS) if priors is None else priors
        self.prior_mu = priors["prior_mu"]
        self.prior_sigma = priors["prior_sigma"]
        self.posterior_mu_initial = priors["posterior_mu_initial"]
        self.posterior_rho_initial = priors["posterior_rho_initial"]
        self.W_mu = Parameter(torch.empty(weight_shape, device=self.device))
        self.W_rho = Parameter(torch.empty(weight_shape, device=self.device))
        if bias:
            self.bias_mu = Parameter(torch.empty(n_out, device=self.device))
            self.bias_rho = Parameter(torch.empty(n_out, device=self.device))
        else:
            self.register_parameter("bias_mu", None)
            self.register_parameter("bias_rho", None)
        self._stream_base = rng.new_stream_base()
        self._kl = None
        self._presampled = None      # (w, bias) handed over by a fused whole-model launch
        self.eps_source = None       # None: on-chip Philox.  callable(shape) -> tensor: external noise (replay tests)
        self.reset_parameters()

    def reset_parameters(self):
        # same draw order as the reference: W_mu, W_rho, bias_mu, bias_rho
        self.W_mu.data.normal_(*self.posterior_mu_initial)
        self.W_rho.data.normal_(*self.posterior_rho_initial)
        if self.use_bias:
            self.bias_mu.data.normal_(*self.posterior_mu_initial)
            self.bias_rho.data.normal_(*self.posterior_rho_initial)

    # -- compat: the reference caches W_sigma / bias_sigma as plain attributes during forward -----------
    # (layers/BBB/BBBConv.py:64,69).  Here they are read-only views of the parameters, computed on every read (nothing in this
    # package reads them; upstream recomputes them every forward too): without autograd the sigma of the fused parameter pass
    # (bbb_reparam_kl_fwd's sigma output: the value every sampled weight of a forward is built from) -- NOT cached: writes
    # through `.data` (reset_parameters, a user's p.data.copy_) do not bump the version counter a cache could be keyed on --
    # with autograd enabled on a trainable rho the differentiable torch expression, as upstream.
    def _sigma_of(self, mu, rho, slot):
        if rho is None:
            return None
        if not rho.is_cuda or (torch.is_grad_enabled() and rho.requires_grad):
            return torch.log1p(torch.exp(rho))
        _, sig, _ = ops.reparam_kl_forward([mu.detach()], [rho.detach()], self.prior_mu, self.prior_sigma, [0], 0, 0, draws=1,
                                           sample=False, want_sigma=True, want_kl=False)
        return sig[0]

    @property
    def W_sigma(self):
        return self._sigma_of(self.W_mu, self.W_rho, "_w_sigma_view")

    @property
    def bias_sigma(self):
        return self._sigma_of(self.bias_mu, self.bias_rho, "_b_sigma_view") if self.use_bias else None

    def __getstate__(self):
        st = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        st = dict(st)
        st.pop("_w_sigma_view", None)
        st.pop("_b_sigma_view", None)
        return st

    def _param_lists(self):
        mus, rhos = [self.W_mu], [self.W_rho]
        ids = [self._stream_base]
        if self.use_bias:
            mus.append(self.bias_mu)
            rhos.append(self.bias_rho)
            ids.append(self._stream_base + 1)
        return mus, rhos, ids

    def kl_loss(self):
        """KL of this layer in the reference's (swapped-argument) form; computed by the fused kernel during
        forward and cached, or on demand if no forward ran since the parameters changed."""
        if self._kl is None:
            mus, rhos, _ = self._param_lists()
            kl, _ = ops.kl_only(mus, rhos, self.prior_mu, self.prior_sigma)
            return kl
        return self._kl

    def _take_kl(self, kl):
        self.__dict__["_kl"] = kl                # (a plain attribute: nn.Module.__setattr__ would spend ~2.5 us on its checks)
