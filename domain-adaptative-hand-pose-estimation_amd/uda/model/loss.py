"""``JointsKLLoss`` (reference ``uda/model/loss.py:115-158``) as one fused row kernel: log-softmax,
target normalisation, KL sum, weighting and the gradient w.r.t. the prediction in a single pass."""
import torch
import torch.nn as nn

from mi355 import ops


class _KLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, eps):
        rows, g = ops.kl_heatmap(pred, target, weight, eps, ctx.needs_input_grad[0])
        ctx.save_for_backward(g)
        return ops.reduce_sum(rows.view(-1), 1.0 / rows.numel()), rows

    @staticmethod
    def backward(ctx, gout, grows):
        g, = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        return ops.scale_by_dev(g, gout.contiguous().float()), None, None, None


class JointsKLLoss(nn.Module):
    """KL Divergence for keypoint detection (RegDA).  ``reduction``: 'mean' | 'none'."""

    def __init__(self, reduction='mean', epsilon=0.):
        super().__init__()
        self.reduction = reduction
        self.epsilon = epsilon

    def forward(self, output, target, target_weight=None):
        loss, rows = _KLFn.apply(output, target.detach(), target_weight, float(self.epsilon))
        if self.reduction == 'mean':
            return loss
        elif self.reduction == 'none':
            return rows.detach().mean(dim=-1)     # forward-only, as no caller differentiates it
