"""``JointsKLLoss`` (reference ``uda/model/loss.py:115-158``) as one fused row kernel: log-softmax,
target normalisation, KL sum, weighting and the gradient w.r.t. the prediction in a single pass."""
import torch
import torch.nn as nn

from mi355 import ops
import mi355 as _rt


class _KLFn(torch.autograd.Function):
    """loss = coeff * mean(rows).  The kernel already writes d loss / d pred; the backward multiplies it by the incoming
    gradient unless that is the shared unit scalar of ``mi355.unit_grad`` (``loss.backward(mi355.unit_grad(loss))``: the
    training step's way of saying "the gradient of the total is 1"), in which case it is handed on as it is."""

    @staticmethod
    def forward(ctx, pred, target, weight, eps, coeff):
        ctx.set_materialize_grads(False)      # `rows` is rarely differentiated: no zero tensor per call for its absent gradient
        rows, g = ops.kl_heatmap(pred, target, weight, eps, ctx.needs_input_grad[0], coeff)
        ctx.save_for_backward(g)
        return ops.reduce_sum(rows.view(-1), float(coeff) / rows.numel()), rows

    @staticmethod
    def backward(ctx, gout, grows):
        g, = ctx.saved_tensors
        if g is None or gout is None:
            return None, None, None, None, None
        if _rt.is_unit_grad(gout):
            return g, None, None, None, None
        return ops.scale_by_dev(g, gout.contiguous().float()), None, None, None, None


class JointsKLLoss(nn.Module):
    """KL Divergence for keypoint detection (RegDA).  ``reduction``: 'mean' | 'none'."""

    def __init__(self, reduction='mean', epsilon=0.):
        super().__init__()
        self.reduction = reduction
        self.epsilon = epsilon

    def forward(self, output, target, target_weight=None, scale=1.0):
        """``scale`` (extension, default 1 = reference): coefficient of this term in the total loss, folded into the kernel
        (the training step passes its 2 / 4 / trade-off factors here instead of multiplying 0-dim tensors)."""
        loss, rows = _KLFn.apply(output, target.detach(), target_weight, float(self.epsilon), float(scale))
        if self.reduction == 'mean':
            return loss
        elif self.reduction == 'none':
            return rows.detach().mean(dim=-1) * float(scale)     # forward-only, as no caller differentiates it
