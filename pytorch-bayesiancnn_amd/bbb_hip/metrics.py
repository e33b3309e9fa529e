"""Host-side mirror of the reference's `metrics` helpers that sit on the training / validation loop
(metrics.py:12-14 ELBO, :23-24 acc, :32-46 get_beta upstream), without the per-step device-to-host synchronisation.

The reference's `acc` converts outputs and targets to numpy on every iteration (main_bayesian.py:60, :84), which stalls the
host until the whole step has drained.  `acc` here returns a 0-dim DEVICE tensor; `AccMeter` accumulates on the device and
synchronises once, when the epoch's mean is read.  `get_beta` is the same schedule (pure Python: it never touched the
device) and `ELBO` the same module."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ELBO(nn.Module):
    """metrics.ELBO (metrics.py:7-14): nll_loss(input, target, 'mean') * train_size + beta * kl."""

    def __init__(self, train_size):
        super().__init__()
        self.train_size = train_size

    def forward(self, input, target, kl, beta):
        assert not target.requires_grad
        return F.nll_loss(input, target, reduction="mean") * self.train_size + beta * kl


def acc(outputs, targets):
    """Fraction of rows whose argmax equals the target -- metrics.acc (metrics.py:23-24) as a 0-dim tensor on the
    outputs' device (no synchronisation; call .item() or float() when the number is needed on the host)."""
    return (outputs.argmax(dim=1) == targets).to(torch.float32).mean()


class AccMeter:
    """Running mean of per-batch accuracies (what the loops do with np.mean(accs)), accumulated on the device."""

    def __init__(self):
        self.total, self.n = None, 0

    def update(self, outputs, targets):
        a = acc(outputs.detach(), targets)
        self.total = a if self.total is None else self.total + a
        self.n += 1

    def mean(self):
        return float(self.total / self.n) if self.n else float("nan")


def get_beta(batch_idx, m, beta_type, epoch=None, num_epochs=None):
    """metrics.get_beta (metrics.py:32-46): the KL weight of mini-batch `batch_idx` of `m`."""
    if type(beta_type) is float:
        return beta_type
    if beta_type == "Blundell":
        return 2 ** (m - (batch_idx + 1)) / (2 ** m - 1)
    if beta_type == "Soenderby":
        if epoch is None or num_epochs is None:
            raise ValueError("Soenderby method requires both epoch and num_epochs to be passed.")
        return min(epoch / (num_epochs // 4), 1)
    if beta_type == "Standard":
        return 1 / m
    return 0
