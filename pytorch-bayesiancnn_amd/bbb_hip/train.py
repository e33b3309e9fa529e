"""Training extension: the inner loop of the reference's train_model
(main_bayesian.py:36-62) on the MI355X path -- Monte-Carlo forward with autograd through the HIP kernels
(bbb_hip.ops), the ELBO of metrics.py:12-24, one multi-tensor Adam launch, and, for data-parallel jobs (one process
per GPU, each with its own shard of the batch), ONE all-reduce of the flattened gradients per step.

Not part of the forward metric; forward-only inference is `bbb_hip.ensemble`.
"""
import torch
import torch.nn.functional as F

from . import _lib, ensemble
from ._lib import AdamSegment, check, cur_stream


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) semantics (amsgrad off, no weight decay) with the update of up to 16
    parameter tensors per HIP launch (`bbb_adam_step`).  State layout matches torch's ('step', 'exp_avg', 'exp_avg_sq'),
    so state_dicts are interchangeable with torch.optim.Adam."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_device(p, p.grad)
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.BBBHipError("FusedAdam needs contiguous parameters and gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append((p, st))
            for step, items in by_step.items():
                for s0 in range(0, len(items), _lib.MAX_SEGMENTS):
                    part = items[s0:s0 + _lib.MAX_SEGMENTS]
                    segs = (AdamSegment * len(part))()
                    for i, (p, st) in enumerate(part):
                        segs[i].param, segs[i].grad = p.data_ptr(), p.grad.data_ptr()
                        segs[i].exp_avg, segs[i].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        segs[i].n = p.numel()
                    dev = part[0][0].device
                    with torch.cuda.device(dev):
                        check(L.bbb_adam_step(segs, len(part), float(group["lr"]), float(group["betas"][0]),
                                              float(group["betas"][1]), float(group["eps"]), step, cur_stream(dev)),
                              "bbb_adam_step")
        return loss


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20):
    """Data-parallel gradient averaging: gradients are packed into flat fp32 buckets (one for a BayesianAlexNet: 20 MB)
    and each bucket is ONE all_reduce (RCCL on GPUs); sum / world, written back in place.  Every rank must call this
    with the same parameter order.  Returns the number of collectives issued."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return 0
    n_coll, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * 4 <= bucket_bytes):
            size += grads[j].numel() * 4
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
        i = j
    return n_coll


def elbo(log_outputs, target, kl, beta, train_size):
    """metrics.ELBO.forward (metrics.py:19-24): nll_loss(mean) * train_size + beta * kl."""
    return F.nll_loss(log_outputs, target, reduction="mean") * train_size + beta * kl


def train_step(net, optimizer, x, target, num_ens, beta, train_size, dp_group=None):
    """One iteration of train_model's batch loop (main_bayesian.py:40-58): zero_grad, num_ens stochastic forwards
    (batched over draws), kl / num_ens, logmeanexp, ELBO, backward, [gradient all-reduce], optimizer.step.
    dp_group: data-parallel process group whose ranks hold different shards of the batch (parameters replicated).
    Returns (loss, log_outputs, kl) detached."""
    optimizer.zero_grad()
    log_outputs, kl = ensemble.mc_forward(net, x, num_ens, kl_mode="mean")
    loss = elbo(log_outputs, target, kl, beta, train_size)
    loss.backward()
    if dp_group is not None:
        allreduce_gradients([p for g in optimizer.param_groups for p in g["params"]], dp_group)
    optimizer.step()
    return loss.detach(), log_outputs.detach(), kl.detach()
