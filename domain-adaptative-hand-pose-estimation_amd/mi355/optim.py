"""FusedSGD: torch.optim.SGD(momentum, weight_decay, nesterov) semantics (reference train1.py:141-148) with one
HIP kernel per parameter group over flat fp32 buffers.

Drop-in for ``torch.optim.SGD`` in the reference's training script: same constructor arguments, same
``param_groups`` / ``state_dict()`` layout (``momentum_buffer`` per parameter), so LambdaLR / MultiStepLR and
the reference checkpoint keys keep working.  Parameters that have never received a gradient (e.g. the unused
``backbone.fc``) are skipped exactly as torch does.
"""
import torch
from torch.optim import Optimizer

import mi355 as _rt
from . import ops
from .nn import mark_grads_fresh, repack_params, repack_params_fp8


class FusedSGD(Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0:
            raise NotImplementedError('dampening is not used by the reference and not built')
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self._flat = None          # list of per-group dicts once flattened
        self._lr_cache = {}

    # ------------------------------------------------------------ flat storage
    @property
    def is_flat(self):
        return self._flat is not None

    def ensure_flat(self):
        """Move every parameter that owns a gradient into flat (param, grad, momentum) buffers.  Parameter
        objects keep their identity; only their storage moves.  Called on the first step()."""
        if self._flat is not None:
            return
        flat = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.requires_grad and p.grad is not None]
            if not ps:
                flat.append(None)
                continue
            dev = ps[0].device
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += (p.numel() + 3) // 4 * 4          # 16-byte aligned slots
            P = torch.zeros(total, dtype=torch.float32, device=dev)
            G = torch.zeros(total, dtype=torch.float32, device=dev)
            M = torch.zeros(total, dtype=torch.float32, device=dev)
            for p, o in zip(ps, offs):
                n = p.numel()
                if p.dtype != torch.float32:
                    raise TypeError('FusedSGD handles fp32 master parameters only')
                pv = P[o:o + n].as_strided(p.shape, p.stride())
                gv = G[o:o + n].as_strided(p.shape, p.stride())
                mv = M[o:o + n].as_strided(p.shape, p.stride())
                pv.copy_(p.data)
                gv.copy_(p.grad)
                st = self.state[p]
                if 'momentum_buffer' in st and st['momentum_buffer'] is not None:
                    mv.copy_(st['momentum_buffer'])
                p.data = pv
                p.grad = gv
                st['momentum_buffer'] = mv
                p._mi_epoch = getattr(p, '_mi_epoch', 0) + 1     # packed copies must be rebuilt (storage moved)
            lr_dev = torch.zeros((), dtype=torch.float32, device=dev)
            flat.append(dict(P=P, G=G, M=M, params=ps, lr_dev=lr_dev, gi=gi,
                             offs={id(p): (o, (p.numel() + 3) // 4 * 4) for p, o in zip(ps, offs)}))
        self._flat = flat
        self.sync_lr(force=True)

    def flat_range(self, params):
        """(G buffer, lo, hi) covering `params` -- which must be neighbours inside one flat group -- or None when the
        optimizer is not flat yet / the parameters own no gradient.  Used to all-reduce a finished part of the backward
        while the rest is still running."""
        if self._flat is None:
            return None
        ids = [id(p) for p in params]
        for f in self._flat:
            if f is None:
                continue
            hit = [f['offs'][i] for i in ids if i in f['offs']]
            if hit:
                lo = min(o for o, _ in hit)
                hi = max(o + n for o, n in hit)
                if sum(n for _, n in hit) != hi - lo:
                    return None                  # not contiguous: leave it to the final pass
                return f['G'], lo, hi
        return None

    def flat_grads(self):
        """The contiguous gradient buffers (one per group) — what the data-parallel all-reduce operates on."""
        _rt.join_side()
        self.ensure_flat()
        return [f['G'] for f in self._flat if f is not None]

    def sync_lr(self, force=False):
        """Push the groups' current learning rates to their device scalars (call outside graph capture)."""
        if self._flat is None:
            return
        for f in self._flat:
            if f is None:
                continue
            lr = float(self.param_groups[f['gi']]['lr'])
            if force or self._lr_cache.get(f['gi']) != lr:
                f['lr_dev'].fill_(lr)
                self._lr_cache[f['gi']] = lr

    # ------------------------------------------------------------ torch.optim API
    def zero_grad(self, set_to_none=False):
        """torch.optim.SGD.zero_grad(set_to_none=True) semantics without freeing anything: a parameter that receives no
        gradient before the next step() is skipped by it (no weight decay, no momentum update), exactly like a parameter
        whose ``grad is None``.  Gradients written by the mi355 kernels are simply overwritten by the next backward (no
        memset, the flat buffers stay in place); gradients accumulated by autograd itself (torch-native modules in the
        same optimizer) are zeroed here and recognised as untouched by their version counter."""
        for group in self.param_groups:
            ps = group['params']
            mark_grads_fresh([p for p in ps if getattr(p, '_mi_slot', False)])
            for p in ps:
                if p.grad is not None and not getattr(p, '_mi_slot', False):
                    p.grad.zero_()
                    p._mi_zero_ver = p.grad._version

    @staticmethod
    def _stale(p):
        """True when `p` received no gradient since the last zero_grad()."""
        if getattr(p, '_mi_slot', False):
            return bool(getattr(p, '_mi_fresh', False))
        return p.grad is not None and getattr(p, '_mi_zero_ver', None) == p.grad._version

    def _late_params(self):
        """Parameters that own a gradient now but did not when the flat buffers were laid out."""
        known = {i for f in self._flat if f is not None for i in f['offs']}
        return [p for g in self.param_groups for p in g['params']
                if p.requires_grad and p.grad is not None and id(p) not in known]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _rt.join_side()              # weight gradients computed on the side stream must have landed
        self.ensure_flat()
        if self._late_params():      # first gradient arrived after the flat layout was fixed: lay it out again
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FusedSGD: a parameter received its first gradient during graph capture; run one eager '
                                   'step with every parameter active before capturing')
            self._flat = None
            self.ensure_flat()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        for f in self._flat:
            if f is None:
                continue
            g = self.param_groups[f['gi']]
            stale = [self._stale(p) for p in f['params']]
            if not any(stale):
                runs = [(0, f['P'].numel())]
            else:                    # step the contiguous runs of parameters that did receive a gradient
                runs, lo = [], None
                for p, st in zip(f['params'], stale):
                    o, n = f['offs'][id(p)]
                    if st:
                        if lo is not None:
                            runs.append((lo, o))
                            lo = None
                    elif lo is None:
                        lo = o
                if lo is not None:
                    runs.append((lo, f['P'].numel()))
            for lo, hi in runs:
                ops.sgd_nesterov(f['P'][lo:hi], f['G'][lo:hi], f['M'][lo:hi], f['lr_dev'], g['momentum'], g['weight_decay'],
                                 g['nesterov'])
            for p, st in zip(f['params'], stale):
                if not st:
                    p._mi_epoch = getattr(p, '_mi_epoch', 0) + 1
            repack_params(f['params'], f.setdefault('pack_cache', {}))     # packed conv copies: one launch per group
            repack_params_fp8(f['params'], f['pack_cache'])
        if _rt.fp8_convs():
            _rt.fp8_tick()           # delayed scaling: the fp8 scales of the next pass from the amax values of this one
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is None:
            return
        for f in self._flat:          # re-alias loaded momentum buffers onto the flat storage
            if f is None:
                continue
            off = 0
            for p in f['params']:
                n = p.numel()
                mv = f['M'][off:off + n].as_strided(p.shape, p.stride())
                st = self.state[p]
                buf = st.get('momentum_buffer')
                if buf is None:
                    mv.zero_()
                elif buf.data_ptr() != mv.data_ptr():
                    mv.copy_(buf)
                st['momentum_buffer'] = mv
                off += (n + 3) // 4 * 4
        self.sync_lr(force=True)
