"""Warm-start gradient layer (reference ``utils/gl.py:8-69``).

Forward is the identity; backward multiplies the gradient by +lambda (the reference multiplies by
``coeff`` with NO sign flip, gl.py:18; lambda in [lo, hi] follows the warm-start schedule of gl.py:59-62).
On this path the layer costs no kernel at all: forward returns an alias of its input tagged with a
device scalar holding lambda, and the dgrad kernels of the layers that consume the alias multiply their
result by that scalar in the epilogue (mi355_conv_dgrad / mi355_pw_k2c ``scale_dev``).
"""
import math
from typing import Optional

import torch
import torch.nn as nn


def warm_start_coeff(iter_num, alpha=1.0, lo=0.0, hi=1.0, max_iters=1000.):
    return float(2.0 * (hi - lo) / (1.0 + math.exp(-alpha * iter_num / max_iters)) - (hi - lo) + lo)


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
        out = input.view(input.shape)          # alias; the tag rides on the Python object
        out._mi_grad_scale = self._coeff_dev
        if self.auto_step:
            self.step()
        return out

    def step(self):
        """Increase iteration number i by 1"""
        self.iter_num += 1
        if self._coeff_dev is not None and not torch.cuda.is_current_stream_capturing():
            self.sync()
