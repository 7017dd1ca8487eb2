"""torch.nn-compatible layers whose forward/backward run on the HIP kernels (through mi355.ops).

They keep torch's parameter names/shapes (state_dict compatible with the reference's nn.Conv2d /
nn.ConvTranspose2d / nn.BatchNorm2d children) but
  * weights live in conv-form memory order [Co][kh][kw][Ci] (a strided view, so `weight.shape` is torch's),
  * feature maps are logical NCHW tensors in channels_last memory, dtype = mi355.compute_dtype(),
  * parameter gradients are written straight into `param.grad` by the wgrad / BN-backward kernels
    (first write after `zero_grad` overwrites, later ones accumulate), not returned through autograd.
"""
import math
import os as _os

import torch
import torch.nn as nn

import mi355 as _rt
from . import Mi355Error, compute_dtype
from . import ops


# ---------------------------------------------------------------- parameter-gradient bookkeeping
def grad_slot(p):
    """Return (tensor to write the gradient into, accumulate?) for parameter `p`."""
    p._mi_slot = True              # this gradient is written by the kernels, not by autograd's AccumulateGrad
    if p.grad is None:
        p.grad = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device)
        p._mi_fresh = False
        return p.grad, False
    if getattr(p, '_mi_fresh', False):
        p._mi_fresh = False
        return p.grad, False
    return p.grad, True


def mark_grads_fresh(params):
    """zero_grad without the memset: the next backward overwrites instead of accumulating."""
    for p in params:
        if p.grad is not None:
            p._mi_fresh = True
    _BWD_PARTIALS.clear()          # leftovers of a backward pass that never reached their BatchNorm
    _LAZY_MASK.clear()
    _BN_DX.clear()
    _GRAD_Q8.clear()               # ... or whose fp8 gradient copy no conv picked up


def _param_version(p):
    return (p._version, getattr(p, '_mi_epoch', 0), p.data_ptr())


# ---- inference: conv -> eval-mode BatchNorm (-> + identity) -> ReLU as ONE launch (test.py's forward-only path)
# A conv / deconv that is followed by a BatchNorm (link_conv_bn) does not run in eval mode under no_grad: it returns a
# _LazyConv; the BatchNorm2d that receives it folds its running statistics into the conv's weights and bias (cached until a
# parameter or a running statistic changes) and launches conv + bias + residual + ReLU (mi355_conv_fwd_act /
# mi355_conv_dgrad_act).  Any other mi355 layer that receives a _LazyConv runs the plain conv first (_as_feature).
_EVAL_FOLD = _os.environ.get('MI355_EVAL_FOLD_BN', '1') == '1'
# 'fp8' mode is a training-throughput configuration: inference (eval-mode modules) takes the BatchNorm-folded bf16 path, which is
# faster (21.8 k vs 19.3 k images/s, ResNet-50 256x256) and more exact than unfolded fp8 convs; 1 = eval-mode convs on fp8 operands too
_FP8_EVAL = _os.environ.get('MI355_FP8_EVAL', '0') == '1'
_BN_GEN = [0]        # bumped by every training-mode BatchNorm forward: its kernels update running statistics in place


class _LazyConv:
    __slots__ = ('conv', 'x')

    def __init__(self, conv, x):
        self.conv, self.x = conv, x

    def materialize(self):
        return self.conv(self.x, _lazy=False)


class _FoldedBn:
    """Weights and bias of one conv with the eval-mode BatchNorm behind it folded in."""

    def __init__(self):
        self.key, self.w, self.bias = None, None, None

    def get(self, conv, bn, dtype, deconv, s2d=False):
        ver = lambda t: None if t is None else _param_version(t)
        key = (ver(conv.weight), ver(conv.bias), ver(bn.weight), ver(bn.bias), bn.running_mean._version, bn.running_var._version,
               _BN_GEN[0], dtype, id(bn), float(bn.eps), s2d)
        if key != self.key:
            with torch.no_grad():
                scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
                shift = bn.bias.detach().float() - bn.running_mean.float() * scale
                if conv.bias is not None:
                    shift = shift + conv.bias.detach().float() * scale
                wm = conv.weight.detach().permute(0, 2, 3, 1)            # memory order [O][kh][kw][I], contiguous view
                if getattr(conv, 'groups', 1) > 1:
                    wm = _dense_from_grouped(conv.weight, conv.groups)   # dense block-diagonal [O][kh][kw][in_channels]
                k2 = conv.kernel_size[0] * conv.kernel_size[1]
                if deconv:       # conv-form (O = in_channels, I = out_channels): the deconv's OUTPUT channels are I
                    wm = (wm * scale.view(1, 1, 1, -1)).contiguous()
                    _, self.w = ops.pack_weights(wm, conv.in_channels, k2, conv.out_channels, conv.out_channels, dtype, want_f=False, want_t=True)
                elif s2d:        # the stem in its folded 4x4 form (Conv2d._s2d_ok)
                    self.w = ops.stem_s2d_pack((wm * scale.view(-1, 1, 1, 1)).contiguous(), dtype)
                else:
                    wm = (wm * scale.view(-1, 1, 1, 1)).contiguous()
                    self.w, _ = ops.pack_weights(wm, conv.out_channels, k2, conv.in_channels, conv._cin_pad(dtype), dtype, want_f=True, want_t=False)
                self.bias = shift.contiguous()
            self.key = key
        return self.w, self.bias


def _lazy_ok(mod, residual=None):
    return (_EVAL_FOLD and mod.bn_follows and not mod.training and not torch.is_grad_enabled() and residual is None and
            getattr(mod, 'mode', 'mfma') == 'mfma' and not (_rt.fp8_convs() and _FP8_EVAL))


def _as_feature(x, dtype):
    """Accept anything NCHW-shaped; hand the kernels a channels_last tensor of the compute dtype."""
    if isinstance(x, _LazyConv):
        x = x.materialize()
    if ops.is_nhwc(x) and x.dtype == dtype:
        return x
    if not x.is_cuda:
        raise Mi355Error('mi355 layers need CUDA/HIP tensors; there is no CPU fallback')
    if x.dtype == torch.float32 and x.is_contiguous():
        return ops.to_nhwc(x, dtype)
    return x.to(dtype).contiguous(memory_format=torch.channels_last)   # foreign layout: torch as glue


# Gradients that still lack a ReLU mask: data_ptr -> (tensor, bit mask).  The last BatchNorm of a residual block hands its
# incoming dy on to the identity branch UNMASKED (no `dresidual` write) with the forward's bit mask registered here; the
# branch's consumer -- conv1's dgrad epilogue (_ConvSkipFn) or the downsample BatchNorm's backward -- applies the mask on the
# fly.  Every other consumer goes through _as_grad, which applies it in place first.  The entry keeps the tensor alive.
_LAZY_MASK = {}
_LAZY_RES = _os.environ.get('MI355_BN_LAZY_DRES', '1') == '1'      # A/B switch
_ZERO_BN_BIAS_GRAD = _os.environ.get('MI355_ZERO_BN_BIAS_GRAD', '1') == '1'      # A/B switch (see _bias_grad)
_STEM_S2D = _os.environ.get('MI355_STEM_S2D', '1') == '1'      # A/B switch: the 7x7 stem as a 4x4 conv over the space-to-depth image
_BN_POOL_FUSE = _os.environ.get('MI355_BN_POOL_FUSE', '1') == '1'      # A/B switch: stem BatchNorm + ReLU + max-pool in one pass


def _take_lazy(g):
    """The pending bit mask of gradient tensor g (None when g is an ordinary gradient); the entry is removed."""
    if not _LAZY_MASK or g is None:
        return None
    ent = _LAZY_MASK.get(g.data_ptr())
    if ent is None or ent[0] is not g:
        return None
    del _LAZY_MASK[g.data_ptr()]
    return ent[1]


def _as_grad(dy, dtype):
    mask = _take_lazy(dy)
    if mask is not None:
        ops.apply_relu_mask(dy, mask)
    if ops.is_nhwc(dy) and dy.dtype == dtype:
        return dy
    return dy.to(dtype).contiguous(memory_format=torch.channels_last)


def _convform_param(Co, Ci, kh, kw):
    """nn.Parameter of torch shape (Co,Ci,kh,kw) stored as [Co][kh][kw][Ci]."""
    return nn.Parameter(torch.empty(Co, kh, kw, Ci).permute(0, 3, 1, 2))


class _PackedWeights:
    """Packed compute-dtype copies of one conv-form weight, refreshed when the master changes."""

    def __init__(self):
        self.key = None
        self.wf = self.wt = None

    def get(self, weight, O, T, I, Ipad, dtype):
        key = (_param_version(weight), dtype, Ipad)
        if key != self.key:
            if self.wf is None or self.wf.dtype != dtype or self.wf.numel() != O * T * Ipad or \
                    self.wf.device != weight.device:
                self.wf = torch.empty(O * T * Ipad, dtype=dtype, device=weight.device)
                self.wt = torch.empty(O * T * Ipad, dtype=dtype, device=weight.device)
            ops.pack_weights_into(weight.detach(), self.wf, self.wt, O, T, I, Ipad, dtype)
            self.key = key
            weight._mi_pack = (self, O, T, I, Ipad)        # lets the optimizer refresh all copies of a group in one launch
        return self.wf, self.wt


def _dense_from_grouped(w, groups):
    """Grouped conv-form weight (Co, Ci/g, kh, kw) -> fp32 dense block-diagonal master [Co][kh][kw][Ci] (zeros off the diagonal)."""
    Co, cig, kh, kw = w.shape
    cog = Co // groups
    wm = w.detach().permute(0, 2, 3, 1).reshape(groups, cog, kh, kw, cig).float()       # conv-form memory: a view
    dense = torch.zeros(groups, cog, kh, kw, groups, cig, dtype=torch.float32, device=w.device)
    idx = torch.arange(groups, device=w.device)
    dense[idx, :, :, :, idx, :] = wm
    return dense.view(Co, kh, kw, groups * cig)


class _PackedGrouped:
    """Forward / input-gradient operands of a grouped conv (ResNeXt): the dense block-diagonal weight, packed like any other.  The
    kernels have no grouped form -- a group of 4 ... 32 channels is far below an MFMA tile -- so the zeros are multiplied too."""

    def __init__(self):
        self.key, self.wf, self.wt = None, None, None

    def get(self, weight, groups, dtype):
        key = (_param_version(weight), dtype)
        if key != self.key:
            Co, cig, kh, kw = weight.shape
            self.wf, self.wt = ops.pack_weights(_dense_from_grouped(weight, groups), Co, kh * kw, groups * cig, groups * cig, dtype)
            self.key = key
        return self.wf, self.wt


class _PackedS2d:
    """Forward operand of the stem in its folded form ([Co][4][4][16], ops.stem_s2d_pack), refreshed when the master changes."""

    def __init__(self):
        self.key, self.wf = None, None

    def get(self, weight, dtype):
        key = (_param_version(weight), dtype)
        if key != self.key:
            if self.wf is None or self.wf.dtype != dtype or self.wf.device != weight.device:
                self.wf = torch.empty(weight.shape[0] * 256, dtype=dtype, device=weight.device)
            ops.stem_s2d_pack(weight.detach(), dtype, out=self.wf)
            self.key = key
        return self.wf


class _PackedFp8:
    """e4m3 copies ([O][T][I] forward operand, [I][T][O] input-gradient operand) of one conv-form weight with their
    per-tensor scale, refreshed when the master changes (just-in-time scale: the weights are small)."""

    def __init__(self):
        self.key = None
        self.wf = self.wt = self.state = self.rec = None

    def get(self, weight, O, T, I):
        key = (_param_version(weight), O, T, I)
        if key != self.key:
            first = self.wf is None or self.wf.numel() != O * T * I or self.wf.device != weight.device
            if first:
                if torch.cuda.is_current_stream_capturing():
                    raise Mi355Error('fp8 weight copies are created on first use: run one eager iteration before capturing')
                self.wf = torch.empty(O * T * I, dtype=torch.uint8, device=weight.device)
                self.wt = torch.empty(O * T * I, dtype=torch.uint8, device=weight.device)
                self.state = _rt.fp8_alloc_state(weight.device, ops.E4M3)
                self.rec = self.state[2:4]          # [1] of this view = state[3]: the descale of the current pack (fp8_common.h)
            # first pack: scale from this tensor; afterwards delayed scaling (weights move by lr per step): ONE launch
            ops.pack_weights_fp8(weight.detach(), O, T, I, self.state, self.wf, self.wt, jit=first)
            self.key = key
            weight._mi_pack8 = (self, O, T, I)          # lets the optimizer refresh all fp8 copies of a group in one launch
        return self.wf, self.wt, self.rec             # consumers descale with the pack's own record, not the live scale


# fp8 copies of GRADIENT tensors written on the side by the kernel that produced them (BatchNorm backward): keyed by address;
# the entry keeps the tensor alive, so the address cannot be reused while the entry exists (same scheme as _BWD_PARTIALS).
_GRAD_Q8 = {}


def _cached_q8(t, fmt):
    """(q8, state) of an fp8 copy that already exists for tensor t, else None."""
    hit = getattr(t, '_mi_q8', None)
    if hit is not None and hit[2] == fmt and hit[3] == t._version:
        return hit[0], hit[1]
    ent = _GRAD_Q8.get(t.data_ptr())
    if ent is not None and ent[3] == fmt and ent[0].shape == t.shape and ent[0].dtype == t.dtype and ent[0]._version == ent[4]:
        return ent[1], ent[2]
    return None


class _Fp8Stream:
    """Scaling state of one fp8 operand stream (a conv's input or output gradient, a BatchNorm's output or input gradient):
    the first tensor is scaled just in time, later ones with the scale derived from the amax of the previous uses
    (mi355.fp8_tick).  A tensor that already carries an fp8 copy (written by its producer, or by another consumer) is not
    quantised again."""

    def __init__(self, fmt):
        self.fmt, self.state = fmt, None

    def ready(self, device):
        return self.state is not None and self.state.device == device

    def quantize(self, t):
        hit = _cached_q8(t, self.fmt)
        if hit is not None:
            return hit
        first = not self.ready(t.device)
        if first:
            if torch.cuda.is_current_stream_capturing():
                raise Mi355Error('fp8 scaling state is created on first use: run one eager iteration before capturing')
            self.state = _rt.fp8_alloc_state(t.device, self.fmt)
        q = ops.fp8_quantize(t, self.state, self.fmt, jit=first)
        t._mi_q8 = (q, q._mi_rec, self.fmt, t._version)       # consumers descale with the copy's own record, not the live state
        return q, q._mi_rec


_PACK_BATCHED = __import__('os').environ.get('MI355_PACK_BATCHED', '1') == '1'


def repack_params(params, cache):
    """Refresh the packed compute-dtype copies of every conv weight in `params` with ONE kernel (called by FusedSGD right
    after its step, instead of one pack launch per conv at the next forward).  `cache`: a dict owned by the caller."""
    if not _PACK_BATCHED:
        return
    ents = [(p, p._mi_pack) for p in params if getattr(p, '_mi_pack', None) is not None and p._mi_pack[0].wf is not None]
    if not ents:
        return
    sig = tuple((p.data_ptr(), e[0].wf.data_ptr(), e[0].wt.data_ptr()) for p, e in ents)
    if cache.get('sig') != sig:
        if torch.cuda.is_current_stream_capturing():
            return                                     # table not built yet: the convs repack lazily, as before
        import numpy as np
        rec = np.zeros(len(ents), dtype=[('w', '<u8'), ('wf', '<u8'), ('wt', '<u8'), ('O', '<i4'), ('T', '<i4'), ('I', '<i4'),
                                         ('Ipad', '<i4'), ('blk0', '<i4'), ('pad', '<i4')])
        blk = 0
        for i, (p, (pk, O, T, I, Ipad)) in enumerate(ents):
            rec[i] = (p.data_ptr(), pk.wf.data_ptr(), pk.wt.data_ptr(), O, T, I, Ipad, blk, 0)
            blk += ((Ipad + 31) // 32) * ((O + 31) // 32) * T
        cache['tab'] = torch.from_numpy(rec.view(np.uint8).copy()).to(ents[0][0].device)
        cache['blocks'], cache['sig'] = blk, sig
    dtype = ents[0][1][0].wf.dtype
    if any(e[0].wf.dtype != dtype for _, e in ents):
        return
    ops.pack_weights_batched(cache['tab'], len(ents), cache['blocks'], dtype)
    for p, (pk, O, T, I, Ipad) in ents:
        pk.key = (_param_version(p), dtype, Ipad)


def repack_params_fp8(params, cache):
    """The fp8 (e4m3) copies of every conv weight in `params` refreshed by ONE kernel right after the optimizer step (scales
    from the amax of the previous pack: delayed scaling), instead of one launch per conv at the next forward."""
    if not _PACK_BATCHED or not _rt.fp8_convs():
        return
    ents = [(p, p._mi_pack8) for p in params if getattr(p, '_mi_pack8', None) is not None and p._mi_pack8[0].wf is not None]
    if not ents:
        return
    sig = tuple((p.data_ptr(), e[0].wf.data_ptr(), e[0].wt.data_ptr(), e[0].state.data_ptr()) for p, e in ents)
    if cache.get('sig8') != sig:
        if torch.cuda.is_current_stream_capturing():
            return                                     # table not built yet: the convs repack lazily
        import numpy as np
        rec = np.zeros(len(ents), dtype=[('w', '<u8'), ('wf', '<u8'), ('wt', '<u8'), ('state', '<u8'), ('O', '<i4'), ('T', '<i4'),
                                         ('I', '<i4'), ('blk0', '<i4')])
        blk = 0
        for i, (p, (pk, O, T, I)) in enumerate(ents):
            rec[i] = (p.data_ptr(), pk.wf.data_ptr(), pk.wt.data_ptr(), pk.state.data_ptr(), O, T, I, blk)
            blk += (O // 32) * (I // 32) * T
        cache['tab8'] = torch.from_numpy(rec.view(np.uint8).copy()).to(ents[0][0].device)
        cache['blocks8'], cache['sig8'] = blk, sig
    ops.pack_weights_fp8_batched(cache['tab8'], len(ents), cache['blocks8'])
    for p, (pk, O, T, I) in ents:
        pk.key = (_param_version(p), O, T, I)


def _chk_convform(weight):
    w = weight.detach()
    if not w.permute(0, 2, 3, 1).is_contiguous():
        raise Mi355Error('conv weight lost its conv-form memory order (was the parameter re-created?)')


# ---------------------------------------------------------------- autograd functions
_FUSE_STATS = _os.environ.get('MI355_BN_STATS_FUSE', '1') == '1'      # A/B switch: BN statistics in the conv epilogue
_FUSE_BNBWD = _os.environ.get('MI355_BN_BWD_FUSE', '0') == '1'        # opt-in: BN backward reduction in the dgrad epilogue (measured neutral)
_SKIP_FUSE = _os.environ.get('MI355_SKIP_FUSE', '1') == '1'             # A/B switch: residual-fork gradient add inside dgrad
_MASK_FROM_Y = _os.environ.get('MI355_BN_MASK_FROM_Y', '0') == '1'     # A/B switch: read y for every ReLU mask
_RELU_BITMASK = _os.environ.get('MI355_BN_RELU_BITMASK', '1') == '1'
# opt-in ('fp8' mode): BatchNorm writes the fp8 copies of y / dx its neighbouring convs consume on the side.  Measured NEGATIVE:
# the extra byte stream costs the BatchNorm kernels +28 .. +80 % (profiles/bn_fp8_side_output_bench.py), more than the stand-alone
# quantisation passes it removes and the fp8 1x1 convs it enables give back (ResNet-50, B=64: 36.0 vs 34.7 ms / iteration)
_FP8_BN_SIDE = _os.environ.get('MI355_FP8_BN_SIDE', '0') == '1'   # A/B switch: bit mask instead of y for BN + residual + ReLU

# dy tensors whose producing GEMM already reduced them for the BatchNorm backward: data_ptr -> (dy, (partial, nslices)).
# The entry keeps dy alive, so its address cannot be reused while the entry exists; BatchNorm's backward pops it.
_BWD_PARTIALS = {}
# input gradients written by a BatchNorm backward whose input came from one of our biased convs (address -> tensor; the entry
# keeps the tensor alive until that conv's backward has looked it up, so the address cannot be reused meanwhile): see _bias_grad
_BN_DX = {}


def _bn_src_of(x):
    """Saved forward state of the training-mode BatchNorm that produced x (None if x is anything else)."""
    return getattr(x, '_mi_bn_src', None) if _FUSE_BNBWD else None


def _dgrad_for_bn(desc, dy, wt, src, x_in, scale_dev=None, out=None, accumulate=False):
    """conv input gradient; when the conv input was a BatchNorm output, reduce it for that BatchNorm on the way out."""
    if src is None:
        return ops.conv_dgrad(desc, dy, wt, scale_dev=scale_dev, out=out, accumulate=accumulate)
    xb, need_y, gamma, beta, mean, invstd, relu = src
    dx, part = ops.conv_dgrad_bnbwd(desc, dy, wt, (xb, x_in if need_y else None, gamma, beta, mean, invstd, relu),
                                    scale_dev=scale_dev, out=out, accumulate=accumulate)
    if part is not None:
        _BWD_PARTIALS[dx.data_ptr()] = (dx, part)
    return dx


def _claim_gl(x, dtype):
    """(autograd input, lambda scalar) for a conv whose input may be the alias returned by WarmStartGradientLayer.
    The conv folds lambda into its own dgrad epilogue, so it takes the layer's INPUT as autograd input (same memory):
    its gradient bypasses the layer's scaling Function.  Anything that is not a ready feature tensor is left alone and
    goes through that Function (correct, one extra kernel)."""
    tag = getattr(x, '_mi_gl', None)
    if tag is None:
        return x, None
    src, scale = tag
    if ops.is_nhwc(src) and src.dtype == dtype and src.data_ptr() == x.data_ptr() and src.shape == x.shape:
        return src, scale
    return x, None


class GradFanIn:
    """Gradient fan-in of a tensor that several mi355 convs consume (the neck output f feeds four heads,
    uda/model/regda_7.py:4931-4946).  Autograd would add the consumers' input gradients with one element-wise kernel per
    extra consumer; instead the first consumer's dgrad writes the buffer and every later one accumulates onto it inside
    its own epilogue (mi355_conv_dgrad accumulate=1) and hands autograd nothing (None = zero): same sum, same bf16
    rounding after each addition, no extra pass over the tensor.  Attach one to the tensor: ``f._mi_fan = GradFanIn()``."""
    __slots__ = ('buf',)

    def __init__(self):
        self.buf = None


def _take_partial(mod, y):
    """Move the statistics partials a conv's forward left on its module onto the output tensor (read by BatchNorm2d)."""
    part = mod._last_partial
    cctx, mod._last_bias_ctx = getattr(mod, '_last_bias_ctx', None), None
    if part is not None:
        mod._last_partial = None
        # the version makes an in-place edit of y before the BatchNorm visible; cctx: the autograd context of a conv WITH a
        # bias -- the BatchNorm that takes these partials tells it that its bias gradient is the column sum of a BatchNorm
        # input gradient, which is zero (see _bias_grad)
        y._mi_bn_partial = (part, y._version, cctx)
    return y


def _zero_grad_once(g):
    """g (the gradient buffer of a bias) <- 0, skipped when this very tensor object still holds the zeros this function wrote last
    time: nothing but this module's bias-gradient code writes non-zeros into it (the optimizer and the gradient exchange read it; a
    mean of zeros is zero; zero_grad only marks it fresh), and a re-created gradient tensor (flattening by FusedSGD, set_to_none)
    is a new object without the mark.  An iteration's ~20 zero-fill launches become none (none is captured into the HIP graphs)."""
    mark = getattr(g, '_mi_zeroed', None)
    # a stand-alone gradient tensor also proves by its version counter that no torch op touched it since; views of FusedSGD's flat
    # buffer share the buffer's counter (every neighbour's zero_ moves it), there the mark alone decides
    if mark is not None and mark is not False and (g._base is not None or mark == g._version):
        return
    g.zero_()
    g._mi_zeroed = g._version


def _bias_grad(ctx, bias, dy):
    """Bias gradient of a conv = column sum of dy.  When the conv's output went straight into a training-mode BatchNorm (which
    consumed the statistics fused into this conv's epilogue and said so: ctx.bias_grad_zero), dy is that BatchNorm's input
    gradient gamma * invstd * (dy_eff - mean(dy_eff) - xhat * mean(dy_eff * xhat)), whose sum over the batch is zero per channel
    in exact arithmetic (sum xhat = 0): the reference's value there is rounding noise around zero; zero is written instead of
    running the reduction (19 of the 23 column sums of an iteration)."""
    g, acc = grad_slot(bias)
    # ... which only holds when dy IS that BatchNorm's input gradient: a conv output that also fed another consumer (a skip
    # connection, an auxiliary loss) receives the sum of several gradients -- a different tensor, not in _BN_DX -> column sum
    ent = _BN_DX.pop(dy.data_ptr(), None)
    from_bn = ent is not None and ent.shape == dy.shape and ent.dtype == dy.dtype
    if getattr(ctx, 'bias_grad_zero', False) and from_bn and _ZERO_BN_BIAS_GRAD:
        if not acc:
            _zero_grad_once(g)
        return
    g._mi_zeroed = False
    ops.colsum(dy, g, acc)


def _conv_forward(ctx, mod, x, bias, residual, stats_ok=True):
    """conv forward on the bf16 / fp32 kernels or, in 'fp8' mode, on fp8 operands; leaves the plan on ctx."""
    want = stats_ok and mod._want_stats() and residual is None
    ok8 = mod._fp8_ok(x)
    ctx.fp8 = ok8 and 'd' in _FP8_PARTS                          # (input gradient on fp8 operands)
    if ok8:
        desc, ctx.desc8, wf8, _, sw = mod._plan_fp8(x)
    if ok8 and 'f' in _FP8_PARTS:
        x8, sx = mod._q_in.quantize(x)
        ctx.x8 = (x8, sx) if mod._fp8_wgrad_ok(x) else None      # the weight gradient reads the same e4m3 copy
        y = ops.conv_fwd_fp8(ctx.desc8, x8, sx, wf8, sw, bias, residual, want_stats=want)
        if want:
            y, mod._last_partial = y
    else:
        desc, wf, _ = mod._plan(x)
        if want:
            y, mod._last_partial = ops.conv_fwd_stats(desc, x, wf, bias)
        else:
            y = ops.conv_fwd(desc, x, wf, bias, residual)
    mod._last_bias_ctx = ctx if (want and bias is not None) else None
    ctx.mod, ctx.desc = mod, desc
    ctx.bn_src = mod._in_bn_src; mod._in_bn_src = None
    return y


# 'fp8' mode: weight gradients of the 3x3 / 4x4 layers from the fp8 copies as well (mi355_conv_wgrad_fp8).  OPT-IN: 2 - 2.7 % faster
# per iteration, but on the synthetic fixed-batch run the supervised loss after 300 / 600 iterations is 29.1 / 24.0 against
# 22.1 / 19.0 with bf16 weight gradients (and 20.6 / 7.0 in bf16 mode): the e5m2 gradient operand costs more than it buys.
_FP8_WGRAD = _os.environ.get('MI355_FP8_WGRAD', '0') == '1'
_FP8_PARTS = _os.environ.get('MI355_FP8_PARTS', 'fd')      # experiment: which GEMMs of the fp8 convs take fp8 operands (f forward, d input gradient)
# 'fp8' mode: the 4x4 transposed convs of the neck on fp8 operands too.  OPT-IN: they are worth 0.2 ms of a 32.4 ms iteration
# (0.6 of 77 at 512x512) and cost most of what 'fp8' mode loses in convergence on the synthetic fixed-batch run (supervised loss
# after 600 iterations: 7.0 in bf16 mode, 10.4 with fp8 convs only, 18.9 with fp8 transposed convs only, 19.0 with both)
_FP8_DECONV = _os.environ.get('MI355_FP8_DECONV', '0') == '1'
_GRAD_FMT = ops.E4M3 if _os.environ.get('MI355_FP8_GRAD_FMT', 'e5m2') == 'e4m3' else ops.E5M2      # experiment: gradient operand format
_FP8_WGRAD_SKIP = _os.environ.get('MI355_FP8_WGRAD_SKIP', '').split(',')      # experiment: 's1', 's2', 'dc' keep their bf16 kernels


def _conv_wgrad(ctx, x, dy, weight):
    """Weight gradient of a conv layer: on the fp8 copies of x (kept from the forward) and dy (made here, reused by the input
    gradient) for the 3x3 / stride-1 layers of 'fp8' mode, on the bf16 / fp32 kernels otherwise."""
    mod = ctx.mod
    x8 = getattr(ctx, 'x8', None)
    if ctx.fp8 and x8 is not None:
        dy8, sdy = mod._q_dy.quantize(dy)
        g, acc = grad_slot(weight)
        ops.conv_wgrad_fp8(ctx.desc8, x8[0], x8[1], dy8, sdy, g, acc, dy_fmt=_GRAD_FMT)
        ctx.x8 = None
        return
    mod._wgrad(ctx.desc, x, dy, weight)


def _conv_dgrad(ctx, x, dy, scale_dev=None, out=None, accumulate=False, acc_mask=None):
    mod = ctx.mod
    if acc_mask is not None:
        if ctx.fp8 or ctx.bn_src is not None or not accumulate:
            ops.apply_relu_mask(out, acc_mask)          # (paths without the masked epilogue: mask first, then accumulate)
        else:
            _, _, wt = mod._plan(x)
            return ops.conv_dgrad_masked_acc(ctx.desc, dy, wt, out, acc_mask, scale_dev=scale_dev)
    if ctx.fp8:
        _, _, _, wt8, sw = mod._plan_fp8(x)
        dy8, sdy = mod._q_dy.quantize(dy)
        return ops.conv_dgrad_fp8(ctx.desc8, dy8, sdy, wt8, sw, scale_dev=scale_dev, out=out, accumulate=accumulate, dy_fmt=_GRAD_FMT)
    _, _, wt = mod._plan(x)
    return _dgrad_for_bn(ctx.desc, dy, wt, ctx.bn_src, x, scale_dev=scale_dev, out=out, accumulate=accumulate)


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, mod, scale_dev, fan=None):
        ctx.fan, ctx.scale_dev = fan, scale_dev
        y = _conv_forward(ctx, mod, x, bias, residual)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        mod, desc = ctx.mod, ctx.desc
        dy = _as_grad(dy, x.dtype)
        dx = None
        if ctx.needs_input_grad[1]:
            _conv_wgrad(ctx, x, dy, weight)
        if ctx.needs_input_grad[0]:
            fan = ctx.fan
            onto = fan is not None and fan.buf is not None and fan.buf.shape == x.shape and fan.buf.dtype == x.dtype
            dx = _conv_dgrad(ctx, x, dy, scale_dev=ctx.scale_dev, out=fan.buf if onto else None, accumulate=onto)
            if onto:
                dx = None
            elif fan is not None:
                fan.buf = dx
        if ctx.has_bias and ctx.needs_input_grad[2]:
            _bias_grad(ctx, bias, dy)
        dres = dy if ctx.needs_input_grad[3] else None
        return dx, None, None, dres, None, None, None


class _ConvSkipFn(torch.autograd.Function):
    """conv(x) together with an alias of x for the parallel branch of a residual block (identity / downsample input).
    In backward the conv's input gradient is added onto the branch's gradient inside the dgrad epilogue, which replaces
    the separate element-wise add autograd would run at the fork (one read+write pass of the block input less).
    Contract: the branch gradient is accumulated IN PLACE, so the branch's consumer must hand back a buffer of its own
    (true for BatchNorm2d's residual input and for conv / deconv input gradients -- the only users in the model)."""

    @staticmethod
    def forward(ctx, x, weight, mod):
        y = _conv_forward(ctx, mod, x, None, None)
        ctx.save_for_backward(x, weight)
        return y, x

    @staticmethod
    def backward(ctx, dy, dskip):
        x, weight = ctx.saved_tensors
        mod, desc = ctx.mod, ctx.desc
        dy = _as_grad(dy, x.dtype)
        if ctx.needs_input_grad[1]:
            _conv_wgrad(ctx, x, dy, weight)
        dx = None
        if ctx.needs_input_grad[0]:
            if dskip is None:
                dx = _conv_dgrad(ctx, x, dy)
            else:       # dskip is a gradient buffer this library produced (BN / conv backward): accumulate in place
                lazy = _take_lazy(dskip) if (ops.is_nhwc(dskip) and dskip.dtype == x.dtype) else None
                dx = _conv_dgrad(ctx, x, dy, out=_as_grad(dskip, x.dtype), accumulate=True, acc_mask=lazy)
        elif dskip is not None:
            _as_grad(dskip, x.dtype)         # (nobody consumes it: just settle a pending mask entry)
        return dx, None, None


class _DeconvFn(torch.autograd.Function):
    """ConvTranspose2d = adjoint of its conv-form: forward is conv_dgrad, input-gradient is conv_fwd."""

    @staticmethod
    def forward(ctx, x, weight, mod):
        ctx.fp8 = mod._fp8_ok(x)
        if ctx.fp8:      # conv-form dgrad with the deconv input as the gathered (e4m3) operand
            desc, desc8, _, wt8, sw = mod._plan_fp8(x)
            x8, sx = mod._q_in.quantize(x)
            ctx.x8 = (x8, sx) if mod._fp8_wgrad_ok(x) else None      # the weight gradient reads the same e4m3 copy
            y = ops.conv_dgrad_fp8(desc8, x8, sx, wt8, sw, want_stats=mod._want_stats(), dy_fmt=ops.E4M3)
            if mod._want_stats():
                y, mod._last_partial = y
            ctx.desc8 = desc8
        else:
            desc, wf, wt = mod._plan(x)
            if mod._want_stats():
                y, mod._last_partial = ops.conv_dgrad_stats(desc, x, wt)
            else:
                y = ops.conv_dgrad(desc, x, wt)
        ctx.mod, ctx.desc = mod, desc
        ctx.bn_src = mod._in_bn_src; mod._in_bn_src = None
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        mod, desc = ctx.mod, ctx.desc
        dy = _as_grad(dy, x.dtype)
        dx = None
        x8 = getattr(ctx, 'x8', None) if ctx.fp8 else None
        if ctx.needs_input_grad[1] and x8 is not None:
            # conv-form roles: its input is this layer's output gradient (e5m2 copy, shared with the input gradient below), its
            # output gradient is this layer's input (the e4m3 copy of the forward)
            g, acc = grad_slot(weight)
            dy8, sdy = mod._q_dy.quantize(dy)
            ops.conv_wgrad_fp8(ctx.desc8, dy8, sdy, x8[0], x8[1], g, acc, dy_fmt=ops.E4M3, x_fmt=_GRAD_FMT)
            ctx.x8 = None
        elif ctx.needs_input_grad[1]:
            g, acc = grad_slot(weight)
            if _rt.grouping_wgrads():
                _rt.group_wgrad(desc, dy, x, g, acc)
            else:
                ops.conv_wgrad(desc, dy, x, g, acc)      # conv-form input = dy, conv-form output = x
        if ctx.needs_input_grad[0] and ctx.fp8:
            _, _, wf8, _, sw = mod._plan_fp8(x)
            dy8, sdy = mod._q_dy.quantize(dy)
            dx = ops.conv_fwd_fp8(ctx.desc8, dy8, sdy, wf8, sw, x_fmt=_GRAD_FMT)
        elif ctx.needs_input_grad[0]:
            _, wf, _ = mod._plan(x)
            if ctx.bn_src is None:
                dx = ops.conv_fwd(desc, dy, wf)
            else:
                xb, need_y, gamma, beta, mean, invstd, relu = ctx.bn_src
                dx, part = ops.conv_fwd_bnbwd(desc, dy, wf, (xb, x if need_y else None, gamma, beta, mean, invstd, relu))
                if part is not None:
                    _BWD_PARTIALS[dx.data_ptr()] = (dx, part)
        return dx, None, None


class _BnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, mod, relu, partial=None, lazy_ok=False, tag_dx=False):
        ctx.tag_dx = bool(tag_dx)
        # 'fp8' mode: the e4m3 copy of y for the fp8 conv that consumes it is written by the apply pass itself (delayed scale);
        # the very first tensor of the stream is scaled just in time by a stand-alone pass instead
        q8 = None
        emit = _FP8_BN_SIDE and _rt.fp8_convs() and x.dtype == torch.bfloat16 and mod.num_features % 128 == 0
        if emit and mod._q_out.ready(x.device):
            q8 = (torch.empty_like(x, dtype=torch.uint8), mod._q_out.state)
        ctx.emit = emit
        # The backward's ReLU mask: recomputed from x when no residual was added; with a residual it is (y > 0), kept as a
        # bit mask the apply pass writes on the side (1/16 of the bytes of y) -- y itself only for the A/B switches
        keep_y = relu and (_MASK_FROM_Y or (residual is not None and (_FUSE_BNBWD or not _RELU_BITMASK)))
        mask = ops.bn_relu_mask(x) if (relu and residual is not None and not keep_y and
                                       (x.requires_grad or gamma.requires_grad)) else None
        _BN_GEN[0] += 1
        y, mean, invstd = ops.bn_train_fwd(x, residual, gamma, beta, mod.running_mean, mod.running_var,
                                           mod.num_batches_tracked, mod.eps, mod.momentum, relu, _rt.bn_stat_updates,
                                           partial=partial, relu_mask=mask, q8=q8)
        mod._last_q8 = q8 if q8 is not None else ('jit' if emit else None)      # attached to the returned tensor by the module
        ctx.relu = relu
        ctx.mask = mask
        ctx.lazy_ok = bool(lazy_ok) and _LAZY_RES and mask is not None
        ctx.mod = mod
        ctx.save_for_backward(x, y if keep_y else None, mean, invstd, gamma, beta)
        mod._last_src = (x, bool(keep_y), gamma, beta, mean, invstd, bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, gamma, beta = ctx.saved_tensors
        # dy of a downsample BatchNorm: the unmasked fork gradient of its residual block + the block's ReLU bit mask
        lazy = _take_lazy(dy) if (not ctx.relu and ctx.mask is None and ops.is_nhwc(dy) and dy.dtype == x.dtype) else None
        dy = _as_grad(dy, x.dtype)
        dg = db = None
        acc = False
        if ctx.needs_input_grad[1]:
            dg, acc = grad_slot(gamma)
        if ctx.needs_input_grad[2]:
            db, acc_b = grad_slot(beta)
            acc = acc_b if dg is None else acc
        ent = _BWD_PARTIALS.pop(dy.data_ptr(), None)       # dy already reduced by the GEMM epilogue that produced it?
        partial = ent[1] if (ent is not None and ent[0].shape == dy.shape and ent[0].dtype == dy.dtype) else None
        q8, stream = None, ctx.mod._q_dx
        emit = ctx.emit and _rt.fp8_convs() and ctx.needs_input_grad[0]
        if emit and stream.ready(x.device):
            q8 = (torch.empty_like(x, dtype=torch.uint8), stream.state)
        hand_on = ctx.lazy_ok and ctx.needs_input_grad[3] and lazy is None and partial is None
        if lazy is not None and partial is None:
            dx, dres = ops.bn_bwd(dy, x, None, gamma, mean, invstd, dg, db, acc, True, False, beta=beta, relu_mask=lazy, q8=q8)
        else:
            if lazy is not None:
                ops.apply_relu_mask(dy, lazy)
            dx, dres = ops.bn_bwd(dy, x, y, gamma, mean, invstd, dg, db, acc, ctx.relu, ctx.needs_input_grad[3] and not hand_on,
                                  beta=beta, partial=partial, relu_mask=ctx.mask, q8=q8)
        if hand_on:            # the identity branch gets dy itself; its consumer applies the bit mask (see _LAZY_MASK)
            dres = dy
            _LAZY_MASK[dy.data_ptr()] = (dy, ctx.mask)
        if ctx.tag_dx and ctx.needs_input_grad[0]:
            _BN_DX[dx.data_ptr()] = dx
        if emit:       # e5m2 copy of dx for the fp8 input-gradient GEMM of the conv in front (found again by address)
            if q8 is None:
                q8 = stream.quantize(dx)
            _GRAD_Q8[dx.data_ptr()] = (dx, q8[0], q8[1], ops.E5M2, dx._version)
        return (dx if ctx.needs_input_grad[0] else None), None, None, dres, None, None, None, None, None


class _BnReluPoolFn(torch.autograd.Function):
    """Stem: BatchNorm2d (training, statistics from the conv epilogue) + ReLU + MaxPool2d(3, 2, 1) in one pass; the normalised
    map is never written.  Backward: the pooled gradient is scattered by the stand-alone max-pool backward, then the one-launch
    BatchNorm backward (ReLU mask recomputed from the conv output)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, partial):
        _BN_GEN[0] += 1
        y, arg, mean, invstd = ops.bn_relu_maxpool_fwd(x, gamma, beta, mod.running_mean, mod.running_var, mod.num_batches_tracked,
                                                       mod.eps, mod.momentum, _rt.bn_stat_updates, partial)
        ctx.in_shape = tuple(x.shape)
        ctx.save_for_backward(x, arg, mean, invstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dpool):
        x, arg, mean, invstd, gamma, beta = ctx.saved_tensors
        dy = ops.maxpool_bwd(_as_grad(dpool, x.dtype), arg, ctx.in_shape)
        dg = db = None
        acc = False
        if ctx.needs_input_grad[1]:
            dg, acc = grad_slot(gamma)
        if ctx.needs_input_grad[2]:
            db, acc_b = grad_slot(beta)
            acc = acc_b if dg is None else acc
        dx, _ = ops.bn_bwd(dy, x, None, gamma, mean, invstd, dg, db, acc, True, False, beta=beta)
        return (dx if ctx.needs_input_grad[0] else None), None, None, None, None


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, arg = ops.maxpool_fwd(x)
        ctx.in_shape = tuple(x.shape)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        arg, = ctx.saved_tensors
        return ops.maxpool_bwd(dy.contiguous(memory_format=torch.channels_last), arg, ctx.in_shape)


_PW_MFMA = _os.environ.get('MI355_PW_MFMA', '1') != '0'      # A/B switch: MFMA forms of the heat-map convs


class _CastCopy:
    """compute-dtype copy of a small fp32 master ([K][C] / [C][K] point-wise weights), refreshed on change.
    transposed=True keeps the [K][C] copy of a [C][K] master (torch glue: the matrix is 21 x 256)."""

    def __init__(self, transposed=False):
        self.key, self.buf = None, None
        self.transposed = transposed

    def get(self, w, dtype):
        if dtype == torch.float32 and not self.transposed:
            return w.detach()
        key = (_param_version(w), dtype)
        if key != self.key:
            if self.transposed:
                w2 = w.detach().reshape(w.shape[0], -1)
                if self.buf is None or self.buf.device != w.device or self.buf.dtype != dtype:
                    self.buf = torch.empty(w2.shape[1], w2.shape[0], dtype=dtype, device=w.device)
                self.buf.copy_(w2.t())
            else:
                if self.buf is None or self.buf.device != w.device:
                    self.buf = torch.empty(w.numel(), dtype=dtype, device=w.device)
                ops.cast_f32(w.detach(), self.buf)
            self.key = key
        return self.buf


def _pw_wgrad_mfma(feat, hm, g, acc, kc_layout, K=None):
    """Weight gradient of a 1x1 conv between NHWC features `feat` [N,C,H,W] and heat-maps `hm` [N,K,H,W] on the
    MFMA wgrad kernel: the heat-map operand is re-laid as NHWC with K padded to 32 channels (16 MB at B=64); a caller that
    kept that 32-channel copy from its forward passes it as `hm` together with the real K."""
    N, C, H, W = feat.shape
    K = hm.shape[1] if K is None else K
    hm32 = ops.to_nhwc(hm, feat.dtype, 32)
    tmp = torch.empty(32 * C, dtype=torch.float32, device=feat.device)
    if kc_layout:      # dW[k][c] = sum_p hm[p][k] * feat[p][c]  : conv-form x=feat (Ci=C), dy=hm32 (Co=32)
        ops.conv_wgrad(ops.make_desc(N, H, W, C, 32, 1, 1, 1, 0, feat.dtype), feat, hm32, tmp, False)
        src = tmp.view(32, C)[:K]
    else:              # dW[c][k] = sum_p feat[p][c] * hm[p][k] : conv-form x=hm32 (Ci=32), dy=feat (Co=C)
        ops.conv_wgrad(ops.make_desc(N, H, W, 32, C, 1, 1, 1, 0, feat.dtype), hm32, feat, tmp, False)
        src = tmp.view(C, 32)[:, :K]
    dst = g.view(src.shape)
    dst.add_(src) if acc else dst.copy_(src)


class _PwC2KFn(torch.autograd.Function):
    """1x1 conv C -> K heat-map (NHWC features in, NCHW fp32 heat-map out)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale_dev, mod):
        K = weight.shape[0]
        if _PW_MFMA:
            y = ops.conv1x1_heatmap(x, mod._cast.get(weight, x.dtype), bias, K)
        else:
            y = ops.pw_c2k(x, weight.detach(), bias, K)
        ctx.scale_dev = scale_dev
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = None
        if ctx.needs_input_grad[0]:     # dx[c] = sum_k dy[k] * w[k][c]  -> k2c with the [K][C] weight
            dx = ops.pw_k2c(dy, weight.detach(), None, x.shape[1], x.dtype, scale_dev=ctx.scale_dev, w_transposed=True)
        if ctx.needs_input_grad[1]:
            g, acc = grad_slot(weight)
            if _PW_MFMA:
                _pw_wgrad_mfma(x, dy, g, acc, True)
            else:
                ops.pw_wgrad(x, dy, g, True, acc)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g, acc = grad_slot(bias)
            ops.hm_rowsum(dy, g, acc)
        return dx, None, None, None, None


class _PwK2CFn(torch.autograd.Function):
    """1x1 conv K heat-map -> C features (+ fused residual add)."""

    @staticmethod
    def forward(ctx, hm, weight, bias, residual, dtype, mod=None):
        C = weight.shape[0]
        if mod is not None and mod._want_stats():
            out, mod._last_partial = ops.pw_k2c_stats(hm, weight.detach(), bias, C, dtype, residual=residual)
            mod._last_bias_ctx = ctx if bias is not None else None
        else:
            out = ops.pw_k2c(hm, weight.detach(), bias, C, dtype, residual=residual)
        ctx.has_bias = bias is not None
        ctx.mod = mod
        ctx.save_for_backward(hm, weight, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        hm, weight, bias = ctx.saved_tensors
        dout = _as_grad(dout, dout.dtype)
        dhm = None
        if ctx.needs_input_grad[0]:     # dhm[k] = sum_c dout[c] * w[c][k] -> c2k with the [C][K] weight
            C, K = weight.shape[0], hm.shape[1]
            if _PW_MFMA and ctx.mod is not None and C >= 64 and (C & (C - 1)) == 0:
                # the heat-map gradient is itself a C -> K 1x1 conv with the transposed weight: MFMA kernel, NCHW fp32 out
                dhm = ops.conv1x1_heatmap(dout, ctx.mod._cast_t.get(weight, dout.dtype), None, K)
            else:
                dhm = ops.pw_c2k(dout, weight.detach(), None, K, w_transposed=True)
        if ctx.needs_input_grad[1]:
            g, acc = grad_slot(weight)
            if _PW_MFMA:
                _pw_wgrad_mfma(dout, hm, g, acc, False)
            else:
                ops.pw_wgrad(dout, hm, g, False, acc)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            _bias_grad(ctx, bias, dout)
        dres = dout if ctx.needs_input_grad[3] else None
        return dhm, None, None, dres, None, None


_CAT = _os.environ.get('MI355_CAT', '1') == '1'      # A/B switch: heat-map conv + feature conv of the fusion heads as one concat-K GEMM


class _ConvCatFn(torch.autograd.Function):
    """``heatmap_conv(heatmap) + feature_conv(feature)`` at the entry of the multiscale-fusion heads (reference
    uda/model/regda_7.py:4573-4581, :4649-4662) as ONE implicit GEMM with K = kh*kw*256 + 32: the 21-channel NCHW fp32 heat-map is
    re-laid once as a 32-channel NHWC operand (the copy the heat-map conv's weight gradient needs anyway) and travels as one more
    K tile of the feature conv (mi355_conv_fwd_cat), with both biases and the BatchNorm statistics in the epilogue.  The 134-MB
    feature-conv output is neither written nor re-read by a separate 21 -> 256 pass.  Backward = the two convs' own backward
    kernels on the shared output gradient."""

    @staticmethod
    def forward(ctx, x, hm, wf, bf, wh, bh, fmod, hmod, scale_dev, fan):
        desc, wpk, _ = fmod._plan(x)
        K = hmod.in_channels
        hm32 = ops.to_nhwc(hm, x.dtype, 32)
        w2, _ = hmod._packed.get(wh, hmod.out_channels, 1, K, 32, x.dtype)
        w2 = w2.view(hmod.out_channels, 32)
        want = hmod._want_stats()
        if want:
            y, part = ops.conv_fwd_cat(desc, x, wpk, bf, hm32, w2, bh, want_stats=True)
        else:
            y, part = ops.conv_fwd_cat(desc, x, wpk, bf, hm32, w2, bh), None
        ctx.mod, ctx.desc, ctx.hmod, ctx.K = fmod, desc, hmod, K
        ctx.fp8, ctx.bn_src = False, None
        ctx.fan, ctx.scale_dev = fan, scale_dev
        ctx.has_bias = bf is not None or bh is not None
        ctx.save_for_backward(x, hm32, wf, bf, wh, bh)
        # the hand-off _take_partial reads: the statistics, and the context whose bias gradients the BatchNorm declares zero
        hmod._last_partial = part
        hmod._last_bias_ctx = ctx if (part is not None and ctx.has_bias) else None
        return y

    @staticmethod
    def backward(ctx, dout):
        x, hm32, wf, bf, wh, bh = ctx.saved_tensors
        fmod, hmod, K = ctx.mod, ctx.hmod, ctx.K
        dout = _as_grad(dout, x.dtype)
        # both bias gradients are the column sum of dout: zero when dout is the input gradient of the BatchNorm that follows
        ent = _BN_DX.pop(dout.data_ptr(), None)
        from_bn = ent is not None and ent.shape == dout.shape and ent.dtype == dout.dtype
        zero = getattr(ctx, 'bias_grad_zero', False) and from_bn and _ZERO_BN_BIAS_GRAD
        for i, b in ((3, bf), (5, bh)):
            if b is not None and ctx.needs_input_grad[i]:
                g, acc = grad_slot(b)
                if zero:
                    if not acc:
                        _zero_grad_once(g)
                else:
                    g._mi_zeroed = False
                    ops.colsum(dout, g, acc)
        if ctx.needs_input_grad[2]:
            _conv_wgrad(ctx, x, dout, wf)
        if ctx.needs_input_grad[4]:
            g, acc = grad_slot(wh)
            _pw_wgrad_mfma(dout, hm32, g, acc, False, K=K)
        dx = dhm = None
        if ctx.needs_input_grad[0]:
            fan = ctx.fan
            onto = fan is not None and fan.buf is not None and fan.buf.shape == x.shape and fan.buf.dtype == x.dtype
            dx = _conv_dgrad(ctx, x, dout, scale_dev=ctx.scale_dev, out=fan.buf if onto else None, accumulate=onto)
            if onto:
                dx = None
            elif fan is not None:
                fan.buf = dx
        if ctx.needs_input_grad[1]:      # dhm[k] = sum_c dout[c] * wh[c][k]: a C -> K 1x1 conv with the transposed pack, NCHW fp32 out
            _, wt = hmod._packed.get(wh, hmod.out_channels, 1, K, 32, dout.dtype)
            dhm = ops.conv1x1_heatmap(dout, wt, None, K)
        return dx, dhm, None, None, None, None, None, None, None, None


# ---------------------------------------------------------------- modules
# Per-forward hand-off slots on the layer modules (plain Python values, rewritten at every forward): nn.Module.__setattr__ walks its
# parameter / buffer / sub-module bookkeeping for every assignment (~2.5 us each, 1600 of them per iteration: 4 ms of host time, the
# eager path's margin over the GPU).  These names bypass it.
_HANDOFF_SLOTS = frozenset(('_last_bias_ctx', '_last_partial', '_in_bn_src', '_last_q8', '_last_src', '_stem_tmp'))


class _FastSlots:
    def __setattr__(self, name, value):
        if name in _HANDOFF_SLOTS:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)


class Conv2d(_FastSlots, nn.Module):
    """nn.Conv2d(in, out, k, stride, padding, bias) on the MFMA implicit-GEMM kernels.
    1x1 convs to / from the K-channel heat-maps (K not a multiple of the 16-byte chunk) take the
    dedicated point-wise kernels and exchange NCHW fp32 heat-maps."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, groups=1):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        if groups < 1 or in_channels % groups or out_channels % groups:
            raise ValueError('in_channels and out_channels must be divisible by groups')
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride, self.padding = (k, k), (stride, stride), (padding, padding)
        self.weight = _convform_param(out_channels, in_channels // groups, k, k)
        self._packed_g = _PackedGrouped() if groups > 1 else None
        self._g_tmp = None
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._packed = _PackedWeights()
        self._packed8, self._q_in, self._q_dy = _PackedFp8(), _Fp8Stream(ops.E4M3), _Fp8Stream(_GRAD_FMT)
        self._cast = _CastCopy()
        self._cast_t = _CastCopy(transposed=True)
        self._folded = _FoldedBn()
        self._stem_tmp = None
        self._last_partial = None
        self._in_bn_src = None
        self.bn_follows = False        # set by link_conv_bn(): the next op is a BatchNorm2d over this conv's output
        # the torchvision stem (7x7 / stride 2 / pad 3 over 3 channels) runs as a 4x4 conv over the 2x2 space-to-depth image
        self._s2d = _STEM_S2D and in_channels == 3 and k == 7 and stride == 2 and padding == 3
        self._packed_s2d = _PackedS2d() if self._s2d else None
        self.reset_parameters()

    def _s2d_ok(self, x):
        return self._s2d and x.shape[1] == 3 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0

    def reset_parameters(self):   # nn.Conv2d defaults
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels // self.groups * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return ('{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, padding={padding}'.format(**self.__dict__) +
                (', groups=%d' % self.groups if self.groups > 1 else ''))

    @property
    def mode(self):
        k = self.kernel_size[0]
        if self.groups > 1:
            return 'mfma'
        if k == 1 and self.stride[0] == 1 and self.out_channels % 8 and self.out_channels <= 32:
            return 'c2k'
        if k == 1 and self.stride[0] == 1 and self.in_channels % 8 and self.in_channels <= 32 and self.in_channels > 3:
            return 'k2c'
        return 'mfma'

    def _cin_pad(self, dtype):
        per = 8 if dtype == torch.bfloat16 else 4
        return ((self.in_channels + per - 1) // per) * per

    def _want_stats(self):
        """Training-mode convs hand the BatchNorm that follows them its statistics partials (fused epilogue)."""
        return _FUSE_STATS and self.training and self.bn_follows

    def _plan(self, x):
        N, C, H, W = x.shape
        k = self.kernel_size[0]
        if self._s2d and C == 16:       # x is the folded image (ops.to_nhwc_s2d)
            _chk_convform(self.weight)
            desc = ops.make_desc(N, H, W, 16, self.out_channels, 4, 4, 1, 2, x.dtype, out_hw=(H, W))
            return desc, self._packed_s2d.get(self.weight, x.dtype), None
        Cp = self._cin_pad(x.dtype)
        if C != Cp:
            raise Mi355Error('conv expects %d (padded) input channels, got %d' % (Cp, C))
        _chk_convform(self.weight)
        desc = ops.make_desc(N, H, W, Cp, self.out_channels, k, k, self.stride[0], self.padding[0], x.dtype)
        if self.groups > 1:
            if Cp != self.in_channels:
                raise Mi355Error('grouped conv: in_channels must be a multiple of the 16-byte chunk')
            wf, wt = self._packed_g.get(self.weight, self.groups, x.dtype)
            return desc, wf, wt
        wf, wt = self._packed.get(self.weight, self.out_channels, k * k, self.in_channels, Cp, x.dtype)
        return desc, wf, wt

    def _fp8_ok(self, x):
        """fp8 operands for this conv?  'fp8' compute mode, a K-heavy kernel (3x3 and up: the 1x1 convs are HBM-bound, an
        extra quantisation pass would cost more than the GEMM gains) and channel counts the fp8 K tile (128) divides."""
        if not (_rt.fp8_convs() and (self.training or _FP8_EVAL) and self.mode == 'mfma' and self.groups == 1 and x.dtype == torch.bfloat16 and
                self.in_channels % 128 == 0 and self.out_channels % 128 == 0):
            return False
        # a 1x1 conv is HBM-bound: worth it only when its producer already wrote the fp8 copy of x (BatchNorm side output)
        return self.kernel_size[0] >= 3 or _cached_q8(x, ops.E4M3) is not None

    def _fp8_wgrad_ok(self, x):
        """weight gradient from the fp8 copies too?  The 3x3 / pad-1 layers the two kernels of mi355_conv_wgrad_fp8 take:
        stride 1 with a power-of-two width >= 8, stride 2 with even extents and a power-of-two output width in [8, 64]."""
        H, W = x.shape[2], x.shape[3]
        if not (_FP8_WGRAD and self.kernel_size[0] == 3 and self.padding[0] == 1 and self.weight.requires_grad):
            return False
        if self.stride[0] == 1:
            return W >= 8 and (W & (W - 1)) == 0 and 's1' not in _FP8_WGRAD_SKIP
        if 's2' in _FP8_WGRAD_SKIP:
            return False
        Wo = W // 2                                               # stride 2: the parity-image kernel
        return self.stride[0] == 2 and H % 2 == 0 and W % 2 == 0 and 8 <= Wo <= 64 and (Wo & (Wo - 1)) == 0

    def _plan_fp8(self, x):
        N, C, H, W = x.shape
        k = self.kernel_size[0]
        _chk_convform(self.weight)
        desc = ops.make_desc(N, H, W, C, self.out_channels, k, k, self.stride[0], self.padding[0], x.dtype)
        desc8 = ops.make_desc_fp8(N, H, W, C, self.out_channels, k, k, self.stride[0], self.padding[0])
        wf8, wt8, sw = self._packed8.get(self.weight, self.out_channels, k * k, self.in_channels)
        return desc, desc8, wf8, wt8, sw

    def _wgrad(self, desc, x, dy, weight):
        g, acc = grad_slot(weight)
        if self.groups > 1:     # dense gradient [Co][kh][kw][Ci], then its block diagonal onto the grouped (Co, Ci/g, kh, kw) gradient
            k, G = self.kernel_size[0], self.groups
            n = self.out_channels * k * k * self.in_channels
            if self._g_tmp is None or self._g_tmp.device != x.device or self._g_tmp.numel() != n:
                self._g_tmp = torch.empty(n, dtype=torch.float32, device=x.device)
            ops.conv_wgrad(desc, x, dy, self._g_tmp, False)
            idx = torch.arange(G, device=x.device)
            diag = self._g_tmp.view(G, self.out_channels // G, k, k, G, self.in_channels // G)[idx, :, :, :, idx, :]
            dst = g.permute(0, 2, 3, 1)
            src = diag.reshape(self.out_channels, k, k, self.in_channels // G)
            dst.add_(src) if acc else dst.copy_(src)
            return
        if desc.Ci == self.in_channels:
            if _rt.grouping_wgrads():
                _rt.group_wgrad(desc, x, dy, g, acc)
            else:
                ops.conv_wgrad(desc, x, dy, g, acc)
        elif desc.kh == 4 and self._s2d:   # folded stem: [Co][4][4][16] gradient, scattered back onto the (Co,3,7,7) one
            if self._stem_tmp is None or self._stem_tmp.device != x.device or self._stem_tmp.numel() != self.out_channels * 256:
                self._stem_tmp = torch.empty(self.out_channels * 256, dtype=torch.float32, device=x.device)
            ops.conv_wgrad(desc, x, dy, self._stem_tmp, False)
            dst = g.permute(0, 2, 3, 1)
            if dst.is_contiguous():
                ops.stem_s2d_unpack_grad(self._stem_tmp, g, acc)
            else:                          # a gradient tensor in a foreign memory order: torch as glue
                t = torch.empty(dst.shape, dtype=torch.float32, device=x.device)
                ops.stem_s2d_unpack_grad(self._stem_tmp, t, False)
                dst.add_(t) if acc else dst.copy_(t)
        else:   # stem: kernel works on the padded channel count; un-pad into the (Co,3,7,7) gradient
            k = self.kernel_size[0]
            if self._stem_tmp is None or self._stem_tmp.device != x.device:
                self._stem_tmp = torch.empty(self.out_channels, k, k, desc.Ci, dtype=torch.float32, device=x.device)
            ops.conv_wgrad(desc, x, dy, self._stem_tmp, False)
            src = self._stem_tmp[..., :self.in_channels]
            dst = g.permute(0, 2, 3, 1)
            dst.add_(src) if acc else dst.copy_(src)

    def forward(self, x, residual=None, _lazy=True):
        if _lazy and _lazy_ok(self, residual) and not isinstance(x, _LazyConv):
            return _LazyConv(self, x)
        dtype = compute_dtype()
        mode = self.mode
        scale = None
        if mode != 'k2c':
            x, scale = _claim_gl(x, dtype)
        if mode == 'c2k':
            x = _as_feature(x, dtype)
            return _PwC2KFn.apply(x, self.weight, self.bias, scale, self)
        if mode == 'k2c':
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.float().contiguous()
            if not x.is_cuda:
                raise Mi355Error('mi355 layers need CUDA/HIP tensors; there is no CPU fallback')
            return _take_partial(self, _PwK2CFn.apply(x, self.weight, self.bias, residual, dtype, self))
        x = self._input_feature(x, dtype)
        self._in_bn_src = _bn_src_of(x)
        fan = getattr(x, '_mi_fan', None) if torch.is_grad_enabled() and x.requires_grad else None
        return _take_partial(self, _ConvFn.apply(x, self.weight, self.bias, residual, self, scale, fan))

    def _input_feature(self, x, dtype):
        if isinstance(x, _LazyConv):
            x = x.materialize()
        if self._s2d_ok(x):
            if not x.is_cuda:
                raise Mi355Error('mi355 layers need CUDA/HIP tensors; there is no CPU fallback')
            return ops.to_nhwc_s2d(x, dtype)
        if x.shape[1] == self.in_channels and self.in_channels != self._cin_pad(dtype):
            return ops.to_nhwc(x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous(), dtype, self._cin_pad(dtype))
        return _as_feature(x, dtype)

    def forward_folded(self, x, bn, residual, relu):
        """act(conv(x) * scale + shift + residual) with bn's running statistics folded into the operands (inference)."""
        dtype = compute_dtype()
        x = self._input_feature(x, dtype)
        N, C, H, W = x.shape
        k = self.kernel_size[0]
        s2d = self._s2d and C == 16
        if s2d:
            desc = ops.make_desc(N, H, W, 16, self.out_channels, 4, 4, 1, 2, x.dtype, out_hw=(H, W))
        else:
            desc = ops.make_desc(N, H, W, C, self.out_channels, k, k, self.stride[0], self.padding[0], x.dtype)
        wf, bias = self._folded.get(self, bn, dtype, False, s2d=s2d)
        if residual is not None:
            residual = _as_feature(residual, dtype)
        return ops.conv_fwd(desc, x, wf, bias, residual, relu=bool(relu))

    def forward_cat(self, x, hm, hconv):
        """hconv(hm) + self(x) as ONE concat-K GEMM (see _ConvCatFn), or None when this call does not qualify (the caller then runs
        the two convs): training mode, this conv on the bf16 / fp32 MFMA kernels with unpadded input channels, hconv a 1x1 conv from
        <= 32 heat-map channels at this conv's output resolution."""
        dtype = compute_dtype()
        k, s_, p_ = self.kernel_size[0], self.stride[0], self.padding[0]
        if not (_CAT and self.training and hconv.training and self.mode == 'mfma' and hconv.mode == 'k2c' and torch.is_tensor(x) and
                torch.is_tensor(hm) and hm.is_cuda and hm.dim() == 4 and x.dim() == 4 and
                hconv.out_channels == self.out_channels and self.out_channels % 8 == 0 and
                self.in_channels == self._cin_pad(dtype) and self.in_channels >= 64 and (self.in_channels & (self.in_channels - 1)) == 0):
            return None
        Ho, Wo = (x.shape[2] + 2 * p_ - k) // s_ + 1, (x.shape[3] + 2 * p_ - k) // s_ + 1
        if tuple(hm.shape) != (x.shape[0], hconv.in_channels, Ho, Wo):
            return None
        x, scale = _claim_gl(x, dtype)
        x = _as_feature(x, dtype)
        if self._fp8_ok(x):
            return None                      # ('fp8' mode: the 3x3 feature conv runs on fp8 operands, which have no concat-K build)
        if hm.dtype != torch.float32 or not hm.is_contiguous():
            hm = hm.float().contiguous()
        fan = getattr(x, '_mi_fan', None) if torch.is_grad_enabled() and x.requires_grad else None
        y = _ConvCatFn.apply(x, hm, self.weight, self.bias, hconv.weight, hconv.bias, self, hconv, scale, fan)
        return _take_partial(hconv, y)

    def forward_skip(self, x):
        """(conv(x), alias of x): for residual blocks, see _ConvSkipFn.  Bias-free MFMA convs only."""
        if _lazy_ok(self):
            return self.forward(x), x
        if not _SKIP_FUSE or self.mode != 'mfma' or self.bias is not None or getattr(x, '_mi_gl', None) is not None or \
                self.in_channels != self._cin_pad(compute_dtype()):
            return self.forward(x), x
        x = _as_feature(x, compute_dtype())
        self._in_bn_src = _bn_src_of(x)
        y, skip = _ConvSkipFn.apply(x, self.weight, self)
        q8 = getattr(x, '_mi_q8', None)
        if q8 is not None and q8[3] == x._version:     # the alias keeps the fp8 copy its producer wrote (down-sample conv)
            skip._mi_q8 = (q8[0], q8[1], q8[2], skip._version)
        return _take_partial(self, y), skip


class ConvTranspose2d(_FastSlots, nn.Module):
    """nn.ConvTranspose2d(in, out, 4, stride=2, padding=1, output_padding=0, bias=False)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=2, padding=1, output_padding=0, bias=False):
        super().__init__()
        if bias or output_padding != 0:
            raise NotImplementedError('only the bias-free, output_padding=0 deconvolution of the pose neck is built')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (kernel_size,) * 2, (stride,) * 2, (padding,) * 2
        # torch shape (in, out, kh, kw) == conv-form (Co=in, Ci=out); memory [in][kh][kw][out]
        self.weight = _convform_param(in_channels, out_channels, kernel_size, kernel_size)
        self.bias = None
        self._packed = _PackedWeights()
        self._folded = _FoldedBn()
        self._packed8, self._q_in, self._q_dy = _PackedFp8(), _Fp8Stream(ops.E4M3), _Fp8Stream(_GRAD_FMT)
        self._last_partial = None
        self._in_bn_src = None
        self.bn_follows = False
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def _fp8_ok(self, x):
        return (_FP8_DECONV and _rt.fp8_convs() and (self.training or _FP8_EVAL) and x.dtype == torch.bfloat16 and
                self.in_channels % 128 == 0 and self.out_channels % 128 == 0)

    def _fp8_wgrad_ok(self, x):
        """weight gradient from the fp8 copies too?  The 4x4 / stride-2 / pad-1 layers whose input width (the conv-form's
        output width) is a power of two in [8, 64] (mi355_conv_wgrad_fp8, parity-image kernel)."""
        W = x.shape[3]
        return (_FP8_WGRAD and 'dc' not in _FP8_WGRAD_SKIP and self.kernel_size[0] == 4 and self.stride[0] == 2 and
                self.padding[0] == 1 and 8 <= W <= 64 and (W & (W - 1)) == 0 and self.weight.requires_grad)

    def _plan_fp8(self, x):
        N, C, H, W = x.shape          # conv-form output side
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        Hi, Wi = (H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k
        _chk_convform(self.weight)
        desc = ops.make_desc(N, Hi, Wi, self.out_channels, self.in_channels, k, k, s, p, x.dtype)
        desc8 = ops.make_desc_fp8(N, Hi, Wi, self.out_channels, self.in_channels, k, k, s, p)
        wf8, wt8, sw = self._packed8.get(self.weight, self.in_channels, k * k, self.out_channels)
        return desc, desc8, wf8, wt8, sw

    def _plan(self, x):
        N, C, H, W = x.shape          # x = conv-form OUTPUT (N, Co=in_channels, Ho, Wo)
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        if C != self.in_channels:
            raise Mi355Error('deconv expects %d input channels, got %d' % (self.in_channels, C))
        Hi, Wi = (H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k
        _chk_convform(self.weight)
        desc = ops.make_desc(N, Hi, Wi, self.out_channels, self.in_channels, k, k, s, p, x.dtype)
        wf, wt = self._packed.get(self.weight, self.in_channels, k * k, self.out_channels, self.out_channels, x.dtype)
        return desc, wf, wt

    def forward(self, x, _lazy=True):
        if _lazy and _lazy_ok(self) and not isinstance(x, _LazyConv):
            return _LazyConv(self, x)
        x = _as_feature(x, compute_dtype())
        self._in_bn_src = _bn_src_of(x)
        return _take_partial(self, _DeconvFn.apply(x, self.weight, self))

    def forward_folded(self, x, bn, residual, relu):
        dtype = compute_dtype()
        x = _as_feature(x, dtype)
        if residual is not None:            # (not a shape of the model: plain deconv, then the BatchNorm's own kernel)
            return ops.bn_eval_fwd(self(x, _lazy=False), _as_feature(residual, dtype), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
        desc, _, _ = self._plan(x)
        wt, bias = self._folded.get(self, bn, dtype, True)
        return ops.deconv_fwd_act(desc, x, wt, bias, relu=bool(relu))

    def _want_stats(self):
        return _FUSE_STATS and self.training and self.bn_follows


class BatchNorm2d(_FastSlots, nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._q_out, self._q_dx = _Fp8Stream(ops.E4M3), _Fp8Stream(ops.E5M2)      # 'fp8' mode: side outputs of y / dx
        self._last_q8 = None

    def extra_repr(self):
        return '{num_features}, eps={eps}, momentum={momentum}'.format(num_features=self.num_features, eps=self.eps, momentum=self.momentum)

    def forward_relu_maxpool(self, x, pool):
        """pool(relu(bn(x))) for the stem: one fused pass in training mode when x carries the statistics of the conv that produced
        it (and the fp8 side outputs are off); the three separate layers otherwise."""
        dtype = compute_dtype()
        tag = None if isinstance(x, _LazyConv) else getattr(x, '_mi_bn_partial', None)
        if (_BN_POOL_FUSE and self.training and tag is not None and ops.is_nhwc(x) and x.dtype == dtype and x._version == tag[1] and
                x.shape[1] == self.num_features and not (_FP8_BN_SIDE and _rt.fp8_convs()) and isinstance(pool, MaxPool2d)):
            return _BnReluPoolFn.apply(x, self.weight, self.bias, self, tag[0])
        return pool(self(x, relu=True))

    def forward(self, x, residual=None, relu=False):
        if isinstance(x, _LazyConv):
            if not self.training and not torch.is_grad_enabled() and x.conv.weight.shape[0 if isinstance(x.conv, Conv2d) else 1] == self.num_features:
                return x.conv.forward_folded(x.x, self, residual, relu)       # conv + this BatchNorm (+ residual, ReLU): one launch
            x = x.materialize()
        x = _as_feature(x, compute_dtype())
        if self.training:
            tag = getattr(x, '_mi_bn_partial', None)          # statistics partials from the conv that produced x
            partial = None
            tag_dx = False
            if tag is not None:
                partial, ver, cctx = tag
                if x._version != ver:
                    raise Mi355Error('the conv output was modified in place before its BatchNorm: the statistics fused into '
                                     'the conv epilogue are stale (use an out-of-place op, or set MI355_BN_STATS_FUSE=0)')
                if x.shape[1] != self.num_features or not ops.is_nhwc(x):
                    partial = None
                elif cctx is not None:
                    cctx.bias_grad_zero = True      # (see _bias_grad)
                    tag_dx = bool(getattr(cctx, 'has_bias', False))
            # the identity branch's gradient may be handed on unmasked when its producer is one of ours that knows how to apply
            # the mask: conv1.forward_skip's alias of the block input, or the downsample BatchNorm
            fn = getattr(residual, 'grad_fn', None) if residual is not None else None
            lazy_ok = fn is not None and type(fn).__name__ in ('_ConvSkipFnBackward', '_BnFnBackward')
            y = _BnFn.apply(x, self.weight, self.bias, residual, self, bool(relu), partial, lazy_ok, tag_dx)
            y._mi_bn_src, self._last_src = self._last_src, None     # lets the consumer conv's dgrad reduce dy for this BN
            q8, self._last_q8 = self._last_q8, None
            if q8 == 'jit':
                self._q_out.quantize(y)                                # first tensor of the stream: scale from its own amax
            elif q8 is not None:
                y._mi_q8 = (q8[0], q8[1], ops.E4M3, y._version)        # fp8 copy written by the apply pass
            return y
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            raise Mi355Error('BatchNorm2d in eval mode is forward-only on this path (wrap it in torch.no_grad())')
        return ops.bn_eval_fwd(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps, relu)


class ReLU(nn.Module):
    """Marker: fused into the preceding BatchNorm2d by FusedSequential / the block forwards."""

    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        raise Mi355Error('a bare ReLU is not built on this path: it is always fused into the BatchNorm before it')


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size=3, stride=2, padding=1):
        super().__init__()
        if (kernel_size, stride, padding) != (3, 2, 1):
            raise NotImplementedError('only the 3x3 s2 p1 stem pool is built')

    def forward(self, x):
        return _MaxPoolFn.apply(_as_feature(x, compute_dtype()))


def link_conv_bn(module):
    """Mark every conv / deconv child that is directly followed (in registration order) by a BatchNorm2d child: in training
    mode such a conv computes the BatchNorm statistics of its output in its epilogue.  Only a hint -- the BatchNorm uses the
    partials solely when they arrive attached to the very tensor it normalises (``y._mi_bn_partial``); a conv output that is
    modified in place before it reaches the BatchNorm would carry stale statistics: BatchNorm2d compares the tensor's
    autograd version with the one recorded by the conv and raises."""
    kids = list(module.children())
    for a, b in zip(kids, kids[1:]):
        if isinstance(a, (Conv2d, ConvTranspose2d)) and isinstance(b, BatchNorm2d):
            a.bn_follows = getattr(a, 'mode', 'mfma') == 'mfma'


class FusedSequential(nn.Sequential):
    """nn.Sequential with the same child indices (state_dict keys) that runs [BatchNorm2d, ReLU] pairs as one
    fused kernel."""

    def __init__(self, *args):
        super().__init__(*args)
        link_conv_bn(self)

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, BatchNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], ReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x
