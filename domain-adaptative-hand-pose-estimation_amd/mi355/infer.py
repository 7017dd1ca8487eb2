"""Forward-only path (``test.py`` / ``validate()``, reference train1.py:495-536): ``model(x)`` of a fixed input shape replayed
from a HIP graph.  Eager, one forward of the pose network is ~120 launches that the host needs longer to enqueue than the GPU
needs to run; replayed it is one graph launch."""
import torch

from . import graph_capture_mode, nn as _nn


class GraphedForward:
    """``y = GraphedForward(model)(x)``: eval-mode, no-grad forwards are captured per input shape after ``warmup`` eager calls
    (they allocate the workspaces and fold the BatchNorms into the convs) and replayed from then on; anything else (training
    mode, gradients enabled) goes to ``model(x)``.  The graphs are dropped when a parameter, a buffer or a running statistic
    changes (training resumed, ``load_state_dict``).  The returned tensor is a copy: it stays valid across calls."""

    def __init__(self, model, warmup=2):
        self.model, self.warmup = model, warmup
        self._graphs, self._seen, self._stamp = {}, {}, None

    def _state_stamp(self):
        v = _nn._BN_GEN[0]
        for t in self.model.parameters():
            v = v * 1000003 + t._version + getattr(t, '_mi_epoch', 0)
        for t in self.model.buffers():
            v = v * 1000003 + t._version
        return v & ((1 << 62) - 1)

    def __call__(self, x):
        if self.model.training or torch.is_grad_enabled() or not x.is_cuda:
            return self.model(x)
        stamp = self._state_stamp()
        if stamp != self._stamp:
            self._graphs.clear(); self._seen.clear(); self._stamp = stamp
        key = (tuple(x.shape), x.dtype, x.device)
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n <= self.warmup:
                return self.model(x)
            sx = x.clone()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, capture_error_mode=graph_capture_mode()):
                sy = self.model(sx)
            ent = self._graphs[key] = (g, sx, sy)
        g, sx, sy = ent
        sx.copy_(x, non_blocking=True)
        g.replay()
        return sy.clone()
