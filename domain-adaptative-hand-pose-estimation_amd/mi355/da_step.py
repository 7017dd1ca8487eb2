"""One domain-adaptation training iteration = steps A, B, C of the reference's ``train()``
(train1.py:371-458), over the MI355X kernels, optionally replayed from captured HIP graphs.

Differences from the reference loop that do not change any result (documented in DESIGN.md):
  * step B stops its backward at the neck output: the backbone/neck gradients it would produce are
    zeroed by ``optimizer_f.zero_grad()`` at the start of step C before anyone reads them (train1.py:440);
  * step C does not compute weight gradients of the adversarial heads: they are zeroed at the start of the
    next step A (train1.py:372-376) before any optimizer reads them;
  * steps B and C run the backbone, the neck and the main head ONCE on the target batch: step B only updates the
    adversarial heads (train1.py:434-436), so the second forward of the reference (train1.py:441) recomputes bit-identical
    features f_t and y_t; step C re-uses them (with their autograd graph) and the BatchNorm layers of that shared part
    apply their running-stat update twice (mi355_bn_train_fwd stat_updates=2), as two forwards would;
  * pseudo-labels, arg-max, KL and PCK stay on the device (the reference round-trips through numpy 12x / iteration).
Data parallelism: one process per GPU; the flat gradient buffers of the optimizers about to step are
all-reduced (mean) over the default process group between the backward and the optimizer kernels.
"""
import os

import torch
import torch.distributed as dist

import mi355 as _rt
from . import ops
from .nn import mark_grads_fresh


# MI355_GRAD_BUCKET_BF16=1 (opt-in): the fp32 gradient ranges travel as bf16 copies -- half the bytes on every xGMI link, one
# cast each way (read 4 + write 2 bytes per element) -- and are averaged in bf16.  Default: fp32, as the reference would.
BF16_BUCKETS = os.environ.get('MI355_GRAD_BUCKET_BF16', '0') == '1'


def _reduce_mean_(t, async_op=False):
    """In-place mean of tensor t over the ranks; returns (work or None, finish) -- finish() after the wait completes the job."""
    avg = dist.get_backend() == 'nccl'
    ws = dist.get_world_size()
    if BF16_BUCKETS and t.dtype == torch.float32:
        low = t.to(torch.bfloat16)
        w = dist.all_reduce(low, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=async_op)

        def finish():
            t.copy_(low)
            if not avg:
                t.div_(ws)
        return w, finish
    w = dist.all_reduce(t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=async_op)
    return w, ((lambda: None) if avg else (lambda: t.div_(ws)))


# MI355_DDP_FORCE_COLLECTIVES=1: issue every collective of the multi-rank path even in a one-rank group -- the only way to run
# the RCCL calls (init, AVG all-reduce of flat gradient ranges, async work handles, broadcast of the conv-form views) on a
# one-GPU box (tests/test_gpu_ddp.py::test_rccl_single_rank_smoke); a mean over one rank is the identity.
_FORCE_COLLECTIVES = os.environ.get('MI355_DDP_FORCE_COLLECTIVES', '0') == '1'


def _distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES)


def _allreduce_mean(bufs):
    if not _distributed():
        return
    for b in bufs:
        _, finish = _reduce_mean_(b)
        finish()


class _OverlapReducer:
    """Gradient mean of one backward pass, started piecewise while the pass is still running.

    Tensor hooks on the outputs of the neck and of the ResNet stages tell when everything *behind* that tensor has finished
    its backward: the flat gradient range of those parameters is then all-reduced asynchronously (the collective's
    stream waits for the kernels enqueued so far and runs beside the rest of the backward).  ``finish()`` reduces whatever
    no hook covered (always the stem / layer1 range, everything in the very first iteration when the optimizers are not
    flat yet) and waits for all collectives.  The result equals one all-reduce per buffer; only the timing differs."""

    def __init__(self, step, keys):
        self.step, self.keys = step, tuple(keys)
        self.works, self.covered = [], {}
        self.avg = dist.get_backend() == 'nccl'

    def _launch(self, G, lo, hi):
        t = G[lo:hi]
        w, finish = _reduce_mean_(t, async_op=True)
        self.works.append((w, finish))
        self.covered.setdefault(G.data_ptr(), []).append((lo, hi))

    def stage_done(self, stage):
        """Called from a gradient hook: the parameters of `stage` (a key of DAStep._stage_params) have their gradients."""
        for key, params in self.step._stage_params.get(stage, ()):
            if key not in self.keys:
                continue
            opt = self.step.opt[key]
            rng = opt.flat_range(params) if hasattr(opt, 'flat_range') else None
            if rng is not None:
                self._launch(*rng)

    def finish(self):
        for G in self.step._grads(self.keys):
            done = sorted(self.covered.get(G.data_ptr(), []))
            pos = 0
            for lo, hi in done + [(G.numel(), G.numel())]:
                if lo > pos:
                    self._launch(G, pos, lo)
                pos = max(pos, hi)
        for w, done in self.works:
            w.wait()
            done()
        self.works = []


def broadcast_module(module, src=0):
    """Rank `src`'s parameters and buffers to every rank.  Conv weights are strided (conv-form) views: the collective
    runs on their dense memory-order view."""
    if not _distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        d = t.data
        if d.dim() == 4 and not d.is_contiguous():
            d = d.permute(0, 2, 3, 1)
        if not d.is_contiguous():
            raise RuntimeError('cannot broadcast a non-dense tensor of shape %s' % (tuple(t.shape),))
        dist.broadcast(d, src)


class DAStep:
    """Holds the static batch buffers and runs A/B/C.  ``optimizers`` = dict with keys f, h, h_adv, h_adv2,
    h_adv3 (FusedSGD or any torch optimizer), ``criteria`` = dict with keys kl, rd (x6), rd2 (x5), rd1 (x1)."""

    def __init__(self, model, optimizers, criteria, trade_off=1.0, skip_discarded=True, track_accuracy=True):
        self.model, self.opt, self.crit = model, optimizers, criteria
        self.trade_off, self.skip, self.track_acc = trade_off, skip_discarded, track_accuracy
        self.graphs = None
        self.out = {}
        self._adv_params = [p for n in ('head_adv', 'head_adv2', 'head_adv3') for p in getattr(model, n).parameters()]
        self._reducer = None
        self._stage_params = {}
        self.overlap = os.environ.get('MI355_OVERLAP_ALLREDUCE', '1') == '1'
        self._install_stage_hooks()

    # ------------------------------------------------------------------ all-reduce overlapped with the backward
    def _install_stage_hooks(self):
        """Forward hooks that put a gradient hook on the output of the neck and of every ResNet stage.  When such a
        gradient arrives, the modules behind that tensor have finished their backward (see _OverlapReducer)."""
        m = self.model
        bb, up = getattr(m, 'backbone', None), getattr(m, 'upsampling', None)
        if bb is None or up is None or not all(hasattr(bb, n) for n in ('layer1', 'layer2', 'layer3', 'layer4', 'maxpool')):
            return
        heads = [(k, list(getattr(m, n).parameters())) for k, n in (('h', 'head'), ('h_adv', 'head_adv'), ('h_adv2', 'head_adv2'),
                                                                    ('h_adv3', 'head_adv3')) if hasattr(m, n)]
        self._stage_params = {                         # gradient of <tensor> ready  ->  these parameters are done
            'neck_out': heads,
            'layer4_out': [('f', list(up.parameters()))],
            'layer3_out': [('f', list(bb.layer4.parameters()))],
            'layer2_out': [('f', list(bb.layer3.parameters()))],
            'layer1_out': [('f', list(bb.layer2.parameters()))],
            'pool_out': [('f', list(bb.layer1.parameters()))],
        }

        def fwd_hook(stage):
            def hook(mod, inp, out):
                if torch.is_tensor(out) and out.requires_grad:
                    out.register_hook(lambda g, s=stage: self._on_stage_grad(s))
            return hook
        for stage, mod in (('neck_out', up), ('layer4_out', bb.layer4), ('layer3_out', bb.layer3), ('layer2_out', bb.layer2),
                           ('layer1_out', bb.layer1), ('pool_out', bb.maxpool)):
            mod.register_forward_hook(fwd_hook(stage))

    def _on_stage_grad(self, stage):
        _rt.flush_grouped_wgrads()       # the weight gradients of the stage just finished: one grouped launch
        if self._reducer is not None:
            self._reducer.stage_done(stage)

    def _overlap_capturable(self):
        """Can the overlapped exchange be CAPTURED into the HIP graphs?  RCCL collectives are stream work and capture like kernels
        (probed on this stack: profiles/tools/rccl_graph_probe.py -- an async all-reduce on RCCL's stream inside torch.cuda.graph, replayed);
        gloo collectives are host work and cannot.  OPT-IN (MI355_DDP_GRAPH_OVERLAP=1): the default keeps the blocking exchange
        between the graphs -- on one xGMI node the two forms are expected within a few tenths of a millisecond of each other
        (DESIGN.md section 6), the captured form has only ever run in a one-rank group (no multi-GPU box in this build's reach), and
        a capture that hangs on real ranks would cost a whole run, where an exception only costs a fallback."""
        return (self.overlap and _distributed() and dist.get_backend() == 'nccl' and
                os.environ.get('MI355_DDP_GRAPH_OVERLAP', '0') == '1')

    def _begin_reduce(self, keys):
        """Arm the overlapped reducer for the backward about to run: more than one rank, launched eagerly -- or being captured with
        a backend whose collectives capture (the all-reduces then sit in the graph on RCCL's stream, beside the rest of the backward)."""
        capturing = torch.cuda.is_current_stream_capturing()
        self._reducer = _OverlapReducer(self, keys) if (self.overlap and _distributed() and
                                                         (not capturing or self._overlap_capturable())) else None
        if self._reducer is not None:
            # collectives will run beside this backward: the one-launch BatchNorm backward needs every CU for its resident
            # blocks (mi355_bn_set_resident), so this pass takes the three-launch form
            self._bn_resident_prev = _rt.load().mi355_bn_set_resident(0)

    def _end_reduce(self, keys):
        r, self._reducer = self._reducer, None
        if r is not None:
            r.finish()
            _rt.load().mi355_bn_set_resident(self._bn_resident_prev)
        else:
            _allreduce_mean(self._grads(keys))

    # ------------------------------------------------------------------ the three steps (fwd+bwd, then update)
    def _fwdbwd_A(self, b):
        m, c = self.model, self.crit
        for o in self.opt.values():
            o.zero_grad()
        with _rt.grouped_wgrads():
            y_s, y_s_adv, y_s_adv2, y_s_adv3, _ = m(b['x_s'])
            # the coefficients of train1.py:384-387 ride inside the loss kernels (scale=), the total's gradient is the unit scalar
            loss_s = c['kl'](y_s, b['label_s'], b['w_s'], scale=2) + \
                c['rd2'](y_s, y_s_adv2, None, b['w_s'], mode='min', scale=4) + \
                c['rd'](y_s, y_s_adv, None, b['w_s'], mode='min', scale=4) + \
                c['rd1'](y_s, y_s_adv3, b['w_s'], mode='min', scale=4)
            loss_s.backward(_rt.unit_grad(loss_s))
        _rt.join_side()
        self.out.update(loss_s=loss_s.detach(), y_s=y_s.detach(), y_s_adv=y_s_adv.detach())

    def _update_A(self):
        for k in ('f', 'h', 'h_adv', 'h_adv2', 'h_adv3'):
            self.opt[k].step()

    def _fwdbwd_B(self, b):
        m, c, to = self.model, self.crit, self.trade_off
        for k in ('h_adv', 'h_adv2', 'h_adv3'):
            self.opt[k].zero_grad()
        if self.skip:
            with _rt.bn_updates(2):               # this forward also stands for step C's (identical) one
                f_t = m.features(b['x_t'])
                y_t = m.head(f_t).detach()        # only ever used detached (pseudo-labels) in B and C
            self._shared = (f_t, y_t)
            y_t_adv, y_t_adv2, y_t_adv3 = m.adv_heads(f_t.detach())
        else:
            y_t, y_t_adv, y_t_adv2, y_t_adv3, _ = m(b['x_t'])
        loss1 = c['rd1'](y_t, y_t_adv3, b['w_t'], mode='max', scale=0.3 * to)
        H = y_t.shape[-1]
        target5 = ops.bilinear_up(y_t_adv3.detach(), H, 0.5)               # 0.5 * up(y_adv3) ...
        target5 = ops.bilinear_up(y_t_adv2.detach(), H, 1.0, out=target5)   # ... + up(y_adv2)   (train1.py:410-424)
        target0 = ops.bilinear_up(y_t_adv3.detach(), H // 2)
        loss2 = c['rd'](y_t, y_t_adv, target5, b['w_t'], mode='max', scale=to)
        loss3 = c['rd2'](y_t, y_t_adv2, target0, b['w_t'], mode='max', scale=0.3 * to)
        loss_gf = loss1 + loss2 + loss3                 # = 0.3 to rd1 + to rd + 0.3 to rd2   (train1.py:426-432)
        with _rt.grouped_wgrads():
            loss_gf.backward(_rt.unit_grad(loss_gf))
        _rt.join_side()
        self.out.update(loss_gf=loss_gf.detach())

    def _update_B(self):
        for k in ('h_adv2', 'h_adv', 'h_adv3'):
            self.opt[k].step()

    def _fwdbwd_C(self, b):
        m, c, to = self.model, self.crit, self.trade_off
        self.opt['f'].zero_grad()
        if self.skip:
            for p in self._adv_params:
                p.requires_grad_(False)
        try:
            if self.skip:
                f_t, y_t = self._shared
                self._shared = None
                y_t_adv, y_t_adv2, y_t_adv3 = m.adv_heads(f_t)
            else:
                y_t, y_t_adv, y_t_adv2, y_t_adv3, _ = m(b['x_t'])
            loss1 = c['rd2'](y_t, y_t_adv2, None, b['w_t'], mode='min', scale=0.3 * to)
            loss2 = c['rd'](y_t, y_t_adv, None, b['w_t'], mode='min', scale=to)
            loss_gt = loss1 + loss2                     # = 0.3 to rd2 + to rd   (train1.py:443-448)
            with _rt.grouped_wgrads():
                loss_gt.backward(_rt.unit_grad(loss_gt))
            _rt.join_side()
        finally:
            if self.skip:
                for p in self._adv_params:
                    p.requires_grad_(True)
        self.out.update(loss_gt=loss_gt.detach(), y_t=y_t.detach(), y_t_adv=y_t_adv.detach())

    def _update_C(self):
        self.opt['f'].step()

    def _accuracy(self, b):
        """Device side of the four accuracy() calls of train1.py:464-475: arg-max coordinates + PCK distances."""
        if not self.track_acc:
            return
        o = self.out
        _, lab_s, _ = ops.argmax2d(b['label_s'])
        has_t = b.get('label_t') is not None
        lab_t = ops.argmax2d(b['label_t'])[1] if has_t else None
        H, W = o['y_s'].shape[-2:]
        from uda.model.regda_4 import cached_centres
        for name, pred, lab in (('s', o['y_s'], lab_s), ('t', o['y_t'], lab_t), ('s_adv', o['y_s_adv'], lab_s),
                                ('t_adv', o['y_t_adv'], lab_t)):
            if lab is None:
                continue
            xy = cached_centres(pred)            # (y_s / y_t: already computed for the pseudo-labels)
            o['pck_' + name] = ops.pck_dists(xy, lab, W / 10.0, H / 10.0)

    def _grads(self, keys):
        bufs = []
        for k in keys:
            o = self.opt[k]
            if hasattr(o, 'flat_grads'):
                bufs += o.flat_grads()
            else:
                bufs += [p.grad for g in o.param_groups for p in g['params'] if p.grad is not None]
        return bufs

    def check_health(self, where=''):
        """Raise if any kernel of the steps run so far reported an in-launch failure (today: a one-launch BatchNorm backward
        that gave up waiting for its blocks and poisoned its gradients).  Synchronises the device."""
        from . import ops
        ops.bn_resident_check(where)

    # ------------------------------------------------------------------ eager iteration
    def run(self, batch):
        """batch: dict x_s, label_s, w_s, x_t, w_t (+ optional label_t for the PCK bookkeeping)."""
        if self.graphs is not None:
            return self.replay(batch)
        from uda.model.regda_4 import _CENTRES
        _CENTRES.clear()
        self.model.train()
        self._begin_reduce(('f', 'h', 'h_adv', 'h_adv2', 'h_adv3'))
        self._fwdbwd_A(batch)
        self._end_reduce(('f', 'h', 'h_adv', 'h_adv2', 'h_adv3'))
        self._update_A()
        self._fwdbwd_B(batch)                    # (heads only: nothing to overlap with)
        _allreduce_mean(self._grads(('h_adv', 'h_adv2', 'h_adv3')))
        self._update_B()
        self._begin_reduce(('f',))
        self._fwdbwd_C(batch)
        self._end_reduce(('f',))
        self._update_C()
        self._accuracy(batch)
        self.model.step()
        return self.out

    # ------------------------------------------------------------------ graph capture / replay
    def capture(self, batch, warmup=3):
        """Capture the iteration as six HIP graphs (fwd+bwd and update of A, B, C) over static copies of
        `batch`; the gradient all-reduces run eagerly between them.  Everything that changes per iteration
        (inputs, learning rates, GL lambda) is read from device memory."""
        self.model.train()
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.run(self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from uda.model.regda_4 import _CENTRES
        _CENTRES.clear()
        pool = torch.cuda.graph_pool_handle()
        KA, KB, KC = ('f', 'h', 'h_adv', 'h_adv2', 'h_adv3'), ('h_adv', 'h_adv2', 'h_adv3'), ('f',)
        # With RCCL the gradient exchange is captured too: steps A and C with the overlapped reducer (its asynchronous all-reduces
        # become a parallel branch of the backward's graph, started where the gradient hooks fire), step B's small exchange behind
        # its backward.  Replay then issues no collective from the host.  Otherwise (one rank, gloo) the exchange stays between graphs.
        self.exchange_captured = self._overlap_capturable()
        if self.exchange_captured:
            def seg_a():
                self._begin_reduce(KA); self._fwdbwd_A(self.static); self._end_reduce(KA)

            def seg_b():
                self._fwdbwd_B(self.static); _allreduce_mean(self._grads(KB))

            def seg_c():
                self._begin_reduce(KC); self._fwdbwd_C(self.static); self._end_reduce(KC)
        else:
            seg_a, seg_b, seg_c = (lambda: self._fwdbwd_A(self.static)), (lambda: self._fwdbwd_B(self.static)), (lambda: self._fwdbwd_C(self.static))
        segs = [seg_a, self._update_A, seg_b, self._update_B, seg_c, lambda: (self._update_C(), self._accuracy(self.static))]
        mode = _rt.graph_capture_mode()        # 'thread_local' beside an RCCL process group (its watchdog thread polls events)
        graphs = []
        for fn in segs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                fn()
            graphs.append(g)
        self.graphs = graphs          # capturing executes nothing: model / optimizer state is unchanged
        return self

    def choose_launch_mode(self, batch, after=None, threshold=0.85):
        """Multi-rank runs.  With RCCL the overlapped gradient exchange is captured into the HIP graphs, so graph replay is chosen
        outright.  With a backend whose collectives cannot be captured (gloo rehearsals): eager launches keep the exchange
        overlapped with the backward, but only pay off while the host can feed the GPU.  Times one eager iteration on the host (enqueue) and on the GPU (enqueue + drain); if enqueueing
        takes (nearly) as long as running it on ANY rank, every rank should replay graphs instead (collectives between the
        graphs).  Runs one real iteration (`after()` is called behind it: scheduler ticks) and returns 'graph' or 'eager' --
        the same answer on every rank (MAX over ranks).  MI355_DDP_GRAPH=1 / 0 forces the answer."""
        import time
        forced = os.environ.get('MI355_DDP_GRAPH')
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        self.run(batch)
        if after is not None:
            after()
        t_host = time.perf_counter() - t_a
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t_a
        r = torch.tensor([t_host / max(t_gpu, 1e-9)], device=next(self.model.parameters()).device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(r, op=dist.ReduceOp.MAX)
        self.host_gpu_ratio = float(r)
        if forced in ('0', '1'):
            return 'graph' if forced == '1' else 'eager'
        if self._overlap_capturable():
            return 'graph'          # RCCL: the overlapped exchange is captured into the graphs -- replay loses nothing and cannot go host-bound
        # (0.85, not ~1: one sample of the host's enqueue time on a node where eight ranks share the cores is an optimistic estimate, and
        #  an eager run that goes host-bound loses more than the blocking exchange between the graphs costs)
        return 'graph' if self.host_gpu_ratio > threshold else 'eager'

    def _host_tick(self):
        for o in self.opt.values():
            if hasattr(o, 'sync_lr'):
                o.sync_lr()
        gl = getattr(self.model, 'gl_layer', None)
        if gl is not None:
            gl.sync()

    def replay(self, batch=None):
        if batch is not None and batch is not self.static:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v, non_blocking=True)
        self._host_tick()
        g = self.graphs
        if getattr(self, 'exchange_captured', False):      # the collectives are nodes of the graphs
            for gr in g:
                gr.replay()
        else:
            g[0].replay()
            _allreduce_mean(self._grads(('f', 'h', 'h_adv', 'h_adv2', 'h_adv3')))
            g[1].replay()
            g[2].replay()
            _allreduce_mean(self._grads(('h_adv', 'h_adv2', 'h_adv3')))
            g[3].replay()
            g[4].replay()
            _allreduce_mean(self._grads(('f',)))
            g[5].replay()
        self.model.step()
        return self.out


def build_training(model, heatmap_size=64, lr=0.01, momentum=0.9, wd=1e-4, lr_gamma=1e-4, lr_decay=0.75,
                   trade_off=1.0, num_keypoints=21, skip_discarded=True, track_accuracy=True):
    """Criteria, the five SGD optimizers and their LambdaLR schedules exactly as train1.py:131-154 builds them.
    Returns (DAStep, optimizers dict, schedulers dict)."""
    from torch.optim.lr_scheduler import LambdaLR
    from uda.model.loss import JointsKLLoss
    from uda.model.regda_4 import PseudoLabelGenerator
    from uda.model.regda_7 import (PseudoLabelGenerator01, PseudoLabelGenerator03, RegressionDisparityx1,
                                   RegressionDisparityx5, RegressionDisparityx6)
    from .optim import FusedSGD
    crit = dict(
        kl=JointsKLLoss(),
        rd=RegressionDisparityx6(PseudoLabelGenerator(num_keypoints, heatmap_size, heatmap_size), JointsKLLoss(epsilon=1e-7)),
        rd2=RegressionDisparityx5(PseudoLabelGenerator03(num_keypoints, heatmap_size // 2, heatmap_size // 2), JointsKLLoss(epsilon=1e-7)),
        rd1=RegressionDisparityx1(PseudoLabelGenerator01(num_keypoints, heatmap_size // 4, heatmap_size // 4), JointsKLLoss(epsilon=1e-7)))
    mk = lambda ps: FusedSGD(ps, lr=0.1, momentum=momentum, weight_decay=wd, nesterov=True)
    opts = dict(
        f=mk([{'params': model.backbone.parameters(), 'lr': 0.1}, {'params': model.upsampling.parameters(), 'lr': 0.1}]),
        h=mk(model.head.parameters()), h_adv=mk(model.head_adv.parameters()),
        h_adv2=mk(model.head_adv2.parameters()), h_adv3=mk(model.head_adv3.parameters()))
    fn = lambda x: lr * (1. + lr_gamma * float(x)) ** (-lr_decay)
    scheds = {k: LambdaLR(o, fn) for k, o in opts.items()}
    return DAStep(model, opts, crit, trade_off, skip_discarded, track_accuracy), opts, scheds
