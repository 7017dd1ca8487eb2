"""Warm-start gradient layer (reference ``utils/gl.py:8-69``).

Forward is the identity; backward multiplies the gradient by +lambda (the reference multiplies by
``coeff`` with NO sign flip, gl.py:18; lambda in [lo, hi] follows the warm-start schedule of gl.py:59-62).
Forward returns an alias of its input (the reference's ``input * 1.0`` copy is not needed).  Backward is safe by
construction: the alias is the output of an autograd Function whose backward multiplies by lambda
(``mi355_scale_feature``, lambda read from a device scalar), so ANY consumer -- a torch op, a ``.clone()``, user code --
gets the reference's ``grad * coeff``.  The ``mi355.nn`` convolutions do better and pay no kernel for it: they recognise
the alias (``_mi_gl`` = (layer input, lambda scalar)), take the layer's *input* as their autograd input and fold lambda
into their own input-gradient epilogue (``mi355_conv_dgrad`` / ``mi355_pw_k2c`` ``scale_dev``); their gradient then never
passes through the Function, so nothing is scaled twice.  An op that drops the Python tag merely loses the fusion.
"""
import math
from typing import Optional

import torch
import torch.nn as nn


def warm_start_coeff(iter_num, alpha=1.0, lo=0.0, hi=1.0, max_iters=1000.):
    return float(2.0 * (hi - lo) / (1.0 + math.exp(-alpha * iter_num / max_iters)) - (hi - lo) + lo)


class _ScaleGradFn(torch.autograd.Function):
    """identity (alias) forward; backward = grad * (*coeff_dev)   (reference GradientFunction, gl.py:8-18)"""

    @staticmethod
    def forward(ctx, x, coeff_dev):
        ctx.coeff_dev = coeff_dev
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from mi355 import ops
        return ops.scale_feature(g, ctx.coeff_dev), None


class WarmStartGradientLayer(nn.Module):
    def __init__(self, alpha: Optional[float] = 1.0, lo: Optional[float] = 0.0, hi: Optional[float] = 1.,
                 max_iters: Optional[int] = 1000., auto_step: Optional[bool] = False):
        super().__init__()
        self.alpha, self.lo, self.hi = alpha, lo, hi
        self.iter_num = 0
        self.max_iters = max_iters
        self.auto_step = auto_step
        self._coeff_dev = None
        self._coeff_iter = None

    @property
    def coeff(self):
        return warm_start_coeff(self.iter_num, self.alpha, self.lo, self.hi, self.max_iters)

    def sync(self, device=None):
        """Write the current lambda to its device scalar (outside graph capture: a captured graph reads it)."""
        if self._coeff_dev is None or (device is not None and self._coeff_dev.device != device):
            if device is None:
                return
            self._coeff_dev = torch.zeros((), dtype=torch.float32, device=device)
            self._coeff_iter = None
        if self._coeff_iter != self.iter_num:
            self._coeff_dev.fill_(self.coeff)
            self._coeff_iter = self.iter_num

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if not torch.cuda.is_current_stream_capturing():
            self.sync(input.device)
        elif self._coeff_dev is None:
            raise RuntimeError('call gl_layer.sync(device) before capturing a graph')
        out = _ScaleGradFn.apply(input, self._coeff_dev)
        out._mi_gl = (input, self._coeff_dev)    # mi355.nn convs claim the scale through this tag (see module docstring)
        if self.auto_step:
            self.step()
        return out

    def step(self):
        """Increase iteration number i by 1"""
        self.iter_num += 1
        if self._coeff_dev is not None and not torch.cuda.is_current_stream_capturing():
            self.sync()
