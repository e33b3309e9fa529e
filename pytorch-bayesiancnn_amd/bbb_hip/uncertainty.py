"""Uncertainty estimation as a second consumer of the Monte-Carlo forwards.

Mirrors uncertainty_estimation.get_uncertainty_per_image / get_uncertainty_per_batch upstream (lines 37-58, 61-102):
same arguments, same (pred, epistemic, aleatoric) numpy results, but the T forwards run as one batched ensemble
launch and the softmax / moment reduction is one HIP kernel instead of a Python loop over samples with numpy dot/diag.
"""
import torch

from . import ensemble, ops, rng


def get_uncertainty_per_image(model, input_image, T=15, normalized=False):
    """The reference repeats the image T times in ONE batch and calls model() once (so with 'bbb' layers all T rows share
    a single weight draw and the epistemic part is ~0; 'lrt' layers decorrelate the rows).  Same semantics here."""
    images = input_image.unsqueeze(0).repeat(T, 1, 1, 1)
    with torch.no_grad():
        net_out, _ = model(images)
        pred, epi, ale = ops.uncertainty(net_out.unsqueeze(1), normalized=normalized)      # T rows = T "draws", B = 1
    return pred[0].cpu().numpy(), epi[0].cpu().numpy(), ale[0].cpu().numpy()


def get_uncertainty_per_batch(model, batch, T=15, normalized=False):
    """T stochastic forwards of the whole batch (fresh weight draws each), reduced per sample: -> [B, C] arrays."""
    with torch.no_grad():
        seed, call0 = rng.next_calls(T)
        logits, _ = ensemble.mc_logits(model, batch.to(next(model.parameters()).device), T, seed, call0)
        pred, epi, ale = ops.uncertainty(logits, normalized=normalized)
    return pred.cpu().numpy(), epi.cpu().numpy(), ale.cpu().numpy()
