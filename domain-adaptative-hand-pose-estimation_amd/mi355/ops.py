"""Thin tensor-level wrappers over the C ABI (one per entry point of include/mi355pose.h).

Tensors are torch CUDA tensors used as device-memory handles.  Feature maps are *logical* NCHW
tensors in channels_last memory (= the NHWC layout the kernels use), dtype bf16 or fp32; heat-maps
are contiguous NCHW fp32.  Nothing here falls back to torch math.
"""
import ctypes

import torch

from . import (BnBwdSrc, ConvDesc, FP8, Mi355Error, WgradItem, call, compute_dtype, dtype_code, load, ptr, stream_ptr, workspace)


# ---------------------------------------------------------------- layout helpers
def nhwc_empty(N, C, H, W, dtype, device):
    return torch.empty((N, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def is_nhwc(x):
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise Mi355Error('mi355 ops need CUDA/HIP tensors (got a %s tensor); there is no CPU fallback' % t.device)


def to_nhwc(x, dtype=None, cpad=None):
    """NCHW fp32 (contiguous) -> channels_last `dtype`, channels zero-padded to `cpad`."""
    dtype = dtype or compute_dtype()
    N, C, H, W = x.shape
    per = 8 if dtype == torch.bfloat16 else 4
    cpad = cpad or ((C + per - 1) // per) * per
    if is_nhwc(x) and x.dtype == dtype and C == cpad:
        return x
    _chk_dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    y = nhwc_empty(N, cpad, H, W, dtype, x.device)
    call('mi355_nchw_to_nhwc', ptr(x), ptr(y), N, C, H, W, cpad, dtype_code(dtype), stream_ptr())
    return y


def to_nchw_f32(x):
    """channels_last bf16/fp32 -> contiguous NCHW fp32."""
    if not is_nhwc(x):
        raise Mi355Error('to_nchw_f32 expects a channels_last tensor')
    _chk_dev(x)
    N, C, H, W = x.shape
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    call('mi355_nhwc_to_nchw', ptr(x), ptr(y), N, C, H, W, dtype_code(x.dtype), stream_ptr())
    return y


def to_nhwc_s2d(x, dtype=None):
    """3-channel NCHW fp32 image (even extents) -> the 2x2 space-to-depth image (N, 16, H/2, W/2), channels_last `dtype`: the
    input of the stem in its folded 4x4 form (mi355_nchw_to_s2d)."""
    dtype = dtype or compute_dtype()
    N, C, H, W = x.shape
    if C != 3 or H % 2 or W % 2:
        raise Mi355Error('to_nhwc_s2d: a 3-channel image with even extents, got %s' % (tuple(x.shape),))
    _chk_dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    y = nhwc_empty(N, 16, H // 2, W // 2, dtype, x.device)
    call('mi355_nchw_to_s2d', ptr(x), ptr(y), N, H, W, dtype_code(dtype), stream_ptr())
    return y


def stem_s2d_pack(w, dtype, out=None):
    """w: fp32 [Co][7][7][3] memory order -> flat `dtype` [Co][4][4][16], the folded stem's forward operand."""
    _chk_dev(w)
    Co = w.numel() // 147
    if out is None:
        out = torch.empty(Co * 256, dtype=dtype, device=w.device)
    call('mi355_stem_s2d_pack', ptr(w), ptr(out), Co, dtype_code(dtype), stream_ptr())
    return out


def stem_s2d_unpack_grad(gs, g, accumulate):
    """gs: fp32 [Co][4][4][16] weight gradient of the folded stem; g: fp32 gradient in [Co][7][7][3] memory order (= / +=)."""
    _chk_dev(gs, g)
    call('mi355_stem_s2d_unpack_grad', ptr(gs), ptr(g), g.numel() // 147, int(bool(accumulate)), stream_ptr())


def make_desc(N, Hi, Wi, Ci, Co, kh, kw, stride, pad, dtype, out_hw=None):
    """out_hw: (Ho, Wo) of a cropped output (unit stride; forward and weight gradient only)."""
    Ho = (Hi + 2 * pad - kh) // stride + 1
    Wo = (Wi + 2 * pad - kw) // stride + 1
    if out_hw is not None:
        Ho, Wo = out_hw
    return ConvDesc(N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, stride, pad, dtype_code(dtype))


# ---------------------------------------------------------------- convolution family
def conv_fwd(desc, x, w, bias=None, residual=None, relu=False):
    _chk_dev(x, w)
    y = nhwc_empty(desc.N, desc.Co, desc.Ho, desc.Wo, x.dtype, x.device)
    if relu:
        call('mi355_conv_fwd_act', ctypes.byref(desc), ptr(x), ptr(w), ptr(bias), ptr(residual), 1, ptr(y), stream_ptr())
    else:
        call('mi355_conv_fwd', ctypes.byref(desc), ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(y), stream_ptr())
    return y


def deconv_fwd_act(desc, x, wT, bias=None, relu=False):
    """Inference ConvTranspose2d forward (conv-form dgrad) + bias + ReLU in one launch."""
    _chk_dev(x, wT)
    y = nhwc_empty(desc.N, desc.Ci, desc.Hi, desc.Wi, x.dtype, x.device)
    call('mi355_conv_dgrad_act', ctypes.byref(desc), ptr(x), ptr(wT), ptr(bias), int(bool(relu)), ptr(y), stream_ptr())
    return y


def _stats_buf(rows, C, device):
    nbytes = load().mi355_conv_stats_bytes(rows, C)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def conv_fwd_stats(desc, x, w, bias=None):
    """conv forward + BatchNorm statistics partials of y from the epilogue.  Returns (y, (partial, nslices) | None)."""
    _chk_dev(x, w)
    y = nhwc_empty(desc.N, desc.Co, desc.Ho, desc.Wo, x.dtype, x.device)
    partial, nbytes = _stats_buf(desc.N * desc.Ho * desc.Wo, desc.Co, x.device)
    ns = ctypes.c_int(0)
    call('mi355_conv_fwd_stats', ctypes.byref(desc), ptr(x), ptr(w), ptr(bias), ptr(y), ptr(partial), nbytes,
         ctypes.byref(ns), stream_ptr())
    return y, ((partial, ns.value) if ns.value > 0 else None)


def conv_fwd_cat(desc, x, w, bias, x2, w2, bias2, want_stats=False):
    """y = conv(x, w) + x2 * w2^T + bias + bias2 as one implicit GEMM (mi355_conv_fwd_cat).  x2: channels_last [N, c2, Ho, Wo] at
    the output resolution, w2: [Co, c2], both in x's dtype.  Returns y, or (y, (partial, nslices) | None) with want_stats."""
    _chk_dev(x, w, x2, w2)
    c2 = x2.shape[1]
    if x2.dtype != x.dtype or w2.dtype != x.dtype or not is_nhwc(x2) or tuple(x2.shape) != (desc.N, c2, desc.Ho, desc.Wo) or \
            tuple(w2.shape) != (desc.Co, c2) or not w2.is_contiguous():
        raise Mi355Error('conv_fwd_cat: second operand pair must be channels_last [N, c2, Ho, Wo] / contiguous [Co, c2] in the dtype of x')
    y = nhwc_empty(desc.N, desc.Co, desc.Ho, desc.Wo, x.dtype, x.device)
    if not want_stats:
        call('mi355_conv_fwd_cat', ctypes.byref(desc), ptr(x), ptr(w), ptr(bias), ptr(x2), ptr(w2), ptr(bias2), c2, ptr(y),
             None, 0, None, stream_ptr())
        return y
    partial, nbytes = _stats_buf(desc.N * desc.Ho * desc.Wo, desc.Co, x.device)
    ns = ctypes.c_int(0)
    call('mi355_conv_fwd_cat', ctypes.byref(desc), ptr(x), ptr(w), ptr(bias), ptr(x2), ptr(w2), ptr(bias2), c2, ptr(y),
         ptr(partial), nbytes, ctypes.byref(ns), stream_ptr())
    return y, ((partial, ns.value) if ns.value > 0 else None)


def conv_dgrad_stats(desc, dy, wT):
    """ConvTranspose2d forward (conv-form dgrad) + BatchNorm statistics partials of its output."""
    _chk_dev(dy, wT)
    dx = nhwc_empty(desc.N, desc.Ci, desc.Hi, desc.Wi, dy.dtype, dy.device)
    partial, nbytes = _stats_buf(desc.N * desc.Hi * desc.Wi, desc.Ci, dy.device)
    ns = ctypes.c_int(0)
    call('mi355_conv_dgrad_stats', ctypes.byref(desc), ptr(dy), ptr(wT), ptr(dx), ptr(partial), nbytes,
         ctypes.byref(ns), stream_ptr())
    return dx, ((partial, ns.value) if ns.value > 0 else None)


def _bn_src(bn):
    """bn = (x, y_or_None, gamma, beta, mean, invstd, relu) -> BnBwdSrc (the tuple keeps the tensors alive)."""
    x, y, gamma, beta, mean, invstd, relu = bn
    return BnBwdSrc(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), int(bool(relu)))


def conv_dgrad_bnbwd(desc, dy, wT, bn, scale_dev=None, out=None, accumulate=False):
    """conv_dgrad whose result is the dy of the BatchNorm described by `bn`; also returns that BatchNorm's backward
    reduction partials (buffer, nslices) from the epilogue, or None when the launch could not fuse them."""
    _chk_dev(dy, wT)
    dx = out if out is not None else nhwc_empty(desc.N, desc.Ci, desc.Hi, desc.Wi, dy.dtype, dy.device)
    partial, nbytes = _stats_buf(desc.N * desc.Hi * desc.Wi, desc.Ci, dy.device)
    ns = ctypes.c_int(0)
    src = _bn_src(bn)
    call('mi355_conv_dgrad_bnbwd', ctypes.byref(desc), ptr(dy), ptr(wT), ptr(scale_dev), int(accumulate), ptr(dx),
         ctypes.byref(src), ptr(partial), nbytes, ctypes.byref(ns), stream_ptr())
    return dx, ((partial, ns.value) if ns.value > 0 else None)


def conv_fwd_bnbwd(desc, x, w, bn):
    """conv_fwd (the input gradient of a ConvTranspose2d) whose result is the dy of the BatchNorm `bn`."""
    _chk_dev(x, w)
    y = nhwc_empty(desc.N, desc.Co, desc.Ho, desc.Wo, x.dtype, x.device)
    partial, nbytes = _stats_buf(desc.N * desc.Ho * desc.Wo, desc.Co, x.device)
    ns = ctypes.c_int(0)
    src = _bn_src(bn)
    call('mi355_conv_fwd_bnbwd', ctypes.byref(desc), ptr(x), ptr(w), ptr(y), ctypes.byref(src), ptr(partial), nbytes,
         ctypes.byref(ns), stream_ptr())
    return y, ((partial, ns.value) if ns.value > 0 else None)


def conv_dgrad(desc, dy, wT, scale_dev=None, out=None, accumulate=False):
    _chk_dev(dy, wT)
    dx = out if out is not None else nhwc_empty(desc.N, desc.Ci, desc.Hi, desc.Wi, dy.dtype, dy.device)
    call('mi355_conv_dgrad', ctypes.byref(desc), ptr(dy), ptr(wT), 0, ptr(scale_dev), int(accumulate), ptr(dx),
         stream_ptr())
    return dx


def conv_dgrad_masked_acc(desc, dy, wT, out, acc_mask, scale_dev=None):
    """out <- conv_dgrad(dy) + (bit of acc_mask set ? out : 0), in place; returns out."""
    _chk_dev(dy, wT, out, acc_mask)
    call('mi355_conv_dgrad_masked_acc', ctypes.byref(desc), ptr(dy), ptr(wT), ptr(scale_dev), ptr(out), ptr(acc_mask), stream_ptr())
    return out


def apply_relu_mask(g, mask):
    """g <- bit of mask set ? g : 0, in place (g channels_last bf16 / fp32, mask from bn_relu_mask); returns g."""
    _chk_dev(g, mask)
    N, C, H, W = g.shape
    call('mi355_apply_relu_mask', ptr(g), ptr(mask), N * H * W, C, dtype_code(g.dtype), stream_ptr())
    return g


def conv_wgrad(desc, x, dy, dw, accumulate, ws_tag='main'):
    """dw: fp32 buffer in [Co][kh][kw][Ci] memory order (Ci = desc.Ci, i.e. padded for the stem)."""
    _chk_dev(x, dy, dw)
    need = load().mi355_conv_wgrad_workspace(ctypes.byref(desc))
    ws = workspace(need, x.device, ws_tag)
    call('mi355_conv_wgrad', ctypes.byref(desc), ptr(x), ptr(dy), ptr(dw), int(accumulate), ptr(ws), ws.numel(),
         stream_ptr())


def conv_wgrad_grouped(items, ws_tag='main'):
    """items: list of (desc, x, dy, dw, accumulate) as for conv_wgrad.  One C call: small problems share launches."""
    n = len(items)
    arr = (WgradItem * n)()
    for i, (desc, x, dy, dw, acc) in enumerate(items):
        _chk_dev(x, dy, dw)
        arr[i].d = desc
        arr[i].x, arr[i].dy, arr[i].dw, arr[i].accumulate = ptr(x), ptr(dy), ptr(dw), int(acc)
    need = load().mi355_conv_wgrad_grouped_workspace(arr, n)
    ws = workspace(need, items[0][1].device, ws_tag)
    call('mi355_conv_wgrad_grouped', arr, n, ptr(ws), ws.numel(), stream_ptr())


def pack_weights(w_master, O, T, I, Ipad, dtype, want_f=True, want_t=True):
    """fp32 master in [O][T][I] memory order -> (wf [O][T][Ipad], wt [Ipad][T][O]) in `dtype`."""
    _chk_dev(w_master)
    dev = w_master.device
    wf = torch.empty(O * T * Ipad, dtype=dtype, device=dev) if want_f else None
    wt = torch.empty(Ipad * T * O, dtype=dtype, device=dev) if want_t else None
    call('mi355_pack_weights', ptr(w_master), ptr(wf), ptr(wt), O, T, I, Ipad, dtype_code(dtype), stream_ptr())
    return wf, wt


def pack_weights_batched(table, nitems, total_blocks, dtype):
    """table: uint8 device tensor holding `nitems` mi355_pack_item records."""
    call('mi355_pack_weights_batched', ptr(table), int(nitems), int(total_blocks), dtype_code(dtype), stream_ptr())


def pack_weights_into(w_master, wf, wt, O, T, I, Ipad, dtype):
    call('mi355_pack_weights', ptr(w_master), ptr(wf), ptr(wt), O, T, I, Ipad, dtype_code(dtype), stream_ptr())


def colsum(dy, out, accumulate):
    """out[C] (=|+=) column sums of the channels_last tensor dy."""
    N, C, H, W = dy.shape
    rows = N * H * W
    ws = workspace(load().mi355_colsum_workspace(rows, C), dy.device)
    call('mi355_colsum', ptr(dy), ptr(out), rows, C, dtype_code(dy.dtype), int(accumulate), ptr(ws), ws.numel(),
         stream_ptr())


# ---------------------------------------------------------------- fp8 operand path (conv forward / input gradient)
E4M3, E5M2 = 0, 1
FP8_MARGIN = 0          # scale = 2^floor(log2(fmt_max / (amax * 2^margin)))


def fp8_state(device):
    """{scale, descale, amax bits, pad} of one per-tensor scaled fp8 operand (device floats)."""
    return torch.zeros(4, dtype=torch.float32, device=device)


def fp8_amax(x, state):
    _chk_dev(x, state)
    call('mi355_fp8_quantize', ptr(x), 0, ptr(state), x.numel(), dtype_code(x.dtype), 0, 0, stream_ptr())


def fp8_update_scale(states, n=1, fmt=E4M3, margin=None):
    call('mi355_fp8_update_scale', ptr(states), int(n), 4, int(fmt), FP8_MARGIN if margin is None else int(margin), stream_ptr())


def fp8_quantize(x, state, fmt=E4M3, jit=False):
    """bf16 / fp32 device tensor (any dense layout) -> uint8 tensor of the same shape and strides holding
    saturate_fmt(x * state[0]).  jit=True: first take amax(|x|) and derive the scale from it (one extra read of x);
    otherwise the scale already in `state` is used and the amax of x is recorded for the next update (delayed scaling).
    The returned tensor's `_mi_rec` is a 4-float record {scale, descale, 0, 0} of THIS copy, usable wherever a state is."""
    _chk_dev(x, state)
    if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
        raise Mi355Error('fp8_quantize needs a dense tensor')
    if x.numel() % 16:
        raise Mi355Error('fp8_quantize needs a multiple of 16 elements, got %d' % x.numel())
    if jit:
        fp8_amax(x, state)
        fp8_update_scale(state, 1, fmt)
    # the copy carries its own {scale, descale, 0, 0} record in 16 bytes behind the data (q._mi_rec): the stream's state is
    # refreshed at every optimizer step, a copy may be consumed after that (weight gradient of a forward shared by two backwards)
    n = x.numel()
    full = torch.empty(n + 16, dtype=torch.uint8, device=x.device)
    q = full.as_strided(x.shape, x.stride())
    call('mi355_fp8_quantize', ptr(x), ptr(full), ptr(state), n, dtype_code(x.dtype), int(fmt), 2, stream_ptr())
    q._mi_rec = full[n:].view(torch.float32)
    return q


def pack_weights_fp8(w_master, O, T, I, state, wf=None, wt=None, jit=True):
    """fp32 master in [O][T][I] memory order -> e4m3 (wf [O][T][I], wt [I][T][O]); jit: per-tensor scale from this very
    tensor (3 launches), else the scale already in `state` (1 launch; the amax is recorded for the next update)."""
    _chk_dev(w_master, state)
    dev = w_master.device
    wf = torch.empty(O * T * I, dtype=torch.uint8, device=dev) if wf is None else wf
    wt = torch.empty(O * T * I, dtype=torch.uint8, device=dev) if wt is None else wt
    call('mi355_pack_weights_fp8', ptr(w_master), ptr(wf), ptr(wt), ptr(state), O, T, I, FP8_MARGIN if jit else -1, stream_ptr())
    return wf, wt


def pack_weights_fp8_batched(table, nitems, total_blocks):
    """table: uint8 device tensor holding `nitems` mi355_pack8_item records."""
    call('mi355_pack_weights_fp8_batched', ptr(table), int(nitems), int(total_blocks), stream_ptr())


def make_desc_fp8(N, Hi, Wi, Ci, Co, kh, kw, stride, pad):
    Ho = (Hi + 2 * pad - kh) // stride + 1
    Wo = (Wi + 2 * pad - kw) // stride + 1
    return ConvDesc(N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, stride, pad, FP8)


def conv_fwd_fp8(desc, x8, x_state, w8, w_state, bias=None, residual=None, want_stats=False, x_fmt=E4M3):
    """y (bf16, channels_last) = conv(x8, w8) * descale_x * descale_w + bias (+ residual); optionally the BatchNorm
    statistics partials of y.  Returns y or (y, (partial, nslices) | None)."""
    _chk_dev(x8, w8)
    y = nhwc_empty(desc.N, desc.Co, desc.Ho, desc.Wo, torch.bfloat16, x8.device)
    partial, nbytes, ns = None, 0, ctypes.c_int(0)
    if want_stats:
        partial, nbytes = _stats_buf(desc.N * desc.Ho * desc.Wo, desc.Co, x8.device)
    call('mi355_conv_fwd_fp8', ctypes.byref(desc), ptr(x8), int(x_fmt), ptr(w8), ptr(x_state[1:2]), ptr(w_state[1:2]), ptr(bias),
         ptr(residual), ptr(y), ptr(partial), nbytes, ctypes.byref(ns), stream_ptr())
    if want_stats:
        return y, ((partial, ns.value) if ns.value > 0 else None)
    return y


def conv_dgrad_fp8(desc, dy8, dy_state, wT8, w_state, scale_dev=None, out=None, accumulate=False, want_stats=False, dy_fmt=E5M2):
    """dx (bf16) = dgrad(dy8, wT8) * descale_dy * descale_w (* *scale_dev) (+ dx when accumulate)."""
    _chk_dev(dy8, wT8)
    dx = out if out is not None else nhwc_empty(desc.N, desc.Ci, desc.Hi, desc.Wi, torch.bfloat16, dy8.device)
    partial, nbytes, ns = None, 0, ctypes.c_int(0)
    if want_stats:
        partial, nbytes = _stats_buf(desc.N * desc.Hi * desc.Wi, desc.Ci, dy8.device)
    call('mi355_conv_dgrad_fp8', ctypes.byref(desc), ptr(dy8), int(dy_fmt), ptr(wT8), ptr(dy_state[1:2]), ptr(w_state[1:2]),
         ptr(scale_dev), int(accumulate), ptr(dx), ptr(partial), nbytes, ctypes.byref(ns), stream_ptr())
    if want_stats:
        return dx, ((partial, ns.value) if ns.value > 0 else None)
    return dx


def conv_wgrad_fp8(desc, x8, x_state, dy8, dy_state, dw, accumulate, dy_fmt=E5M2, x_fmt=E4M3, ws_tag='main'):
    """dw (fp32, [Co][kh][kw][Ci]) (+)= wgrad(x8, dy8) * descale_x * descale_dy for the 3x3 / stride-1 and the 3x3 / 4x4 /
    stride-2 layers: both operands are the fp8 copies the forward / input-gradient launches of the layer already made."""
    _chk_dev(x8, dy8, dw)
    need = load().mi355_conv_wgrad_fp8_workspace(ctypes.byref(desc))
    ws = workspace(need, x8.device, ws_tag)
    call('mi355_conv_wgrad_fp8', ctypes.byref(desc), ptr(x8), int(x_fmt), ptr(dy8), int(dy_fmt), ptr(x_state[1:2]),
         ptr(dy_state[1:2]), ptr(dw), int(accumulate), ptr(ws), ws.numel(), stream_ptr())


# ---------------------------------------------------------------- batch norm
def bn_relu_mask(x):
    """uint8 buffer for the ReLU bit mask of a BatchNorm over x: one byte per 16-byte chunk of every row."""
    per = 8 if x.dtype == torch.bfloat16 else 4
    return torch.empty(x.numel() // per, dtype=torch.uint8, device=x.device)


def bn_train_fwd(x, residual, gamma, beta, running_mean, running_var, nbt, eps, momentum, relu, stat_updates=1,
                 partial=None, relu_mask=None, q8=None):
    """partial: (buffer, nslices) from conv_fwd_stats / conv_dgrad_stats of the conv that produced x -> no statistics pass.
    relu_mask: bn_relu_mask(x) buffer that receives the (y > 0) bits for the backward (instead of keeping y).
    q8: (uint8 tensor shaped like y, fp8 state) -> also the e4m3 copy of y, scaled by state[0], and its amax."""
    q8_out, q8_state = q8 if q8 is not None else (None, None)
    _chk_dev(x, gamma)
    N, C, H, W = x.shape
    rows = N * H * W
    y = nhwc_empty(N, C, H, W, x.dtype, x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    if partial is not None:
        buf, ns = partial
        ss = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        call('mi355_bn_train_fwd_partials', ptr(x), ptr(residual), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean),
             ptr(running_var), ptr(nbt), ptr(mean), ptr(invstd), rows, C, float(eps), float(momentum), int(stat_updates),
             int(relu), dtype_code(x.dtype), ptr(buf), int(ns), ptr(ss), ptr(relu_mask), ptr(q8_out), ptr(q8_state), stream_ptr())
        return y, mean, invstd
    ws = workspace(load().mi355_bn_workspace(rows, C), x.device)
    call('mi355_bn_train_fwd', ptr(x), ptr(residual), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean),
         ptr(running_var), ptr(nbt), ptr(mean), ptr(invstd), rows, C, float(eps), float(momentum), int(stat_updates), int(relu),
         dtype_code(x.dtype), ptr(ws), ws.numel(), ptr(relu_mask), ptr(q8_out), ptr(q8_state), stream_ptr())
    return y, mean, invstd


def bn_relu_maxpool_fwd(x, gamma, beta, running_mean, running_var, nbt, eps, momentum, stat_updates, partial):
    """BatchNorm (train, statistics partials of the producing conv) + ReLU + MaxPool2d(3, 2, 1) in one pass; returns
    (y_pool, argidx, mean, invstd)."""
    _chk_dev(x, gamma)
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = nhwc_empty(N, C, Ho, Wo, x.dtype, x.device)
    arg = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ss = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    buf, ns = partial
    call('mi355_bn_relu_maxpool_fwd_partials', ptr(x), ptr(y), ptr(arg), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
         ptr(nbt), ptr(mean), ptr(invstd), N, H, W, C, float(eps), float(momentum), int(stat_updates), dtype_code(x.dtype), ptr(buf),
         int(ns), ptr(ss), stream_ptr())
    return y, arg, mean, invstd


def bn_eval_fwd(x, residual, gamma, beta, running_mean, running_var, eps, relu):
    _chk_dev(x, gamma)
    N, C, H, W = x.shape
    y = nhwc_empty(N, C, H, W, x.dtype, x.device)
    call('mi355_bn_eval_fwd', ptr(x), ptr(residual), ptr(y), ptr(gamma), ptr(beta), ptr(running_mean),
         ptr(running_var), N * H * W, C, float(eps), int(relu), dtype_code(x.dtype), stream_ptr())
    return y


def bn_bwd(dy, x, y, gamma, mean, invstd, dgamma, dbeta, accumulate, relu, want_dres, beta=None, partial=None, relu_mask=None,
           q8=None):
    """partial: (buffer, nslices) reduction partials from the GEMM epilogue that produced dy -> no reduction pass.
    relu_mask: the bit mask the forward wrote (then y is not needed).
    q8: (uint8 tensor shaped like dx, fp8 state) -> also the e5m2 copy of dx, scaled by state[0], and its amax."""
    q8_out, q8_state = q8 if q8 is not None else (None, None)
    N, C, H, W = x.shape
    rows = N * H * W
    dx = nhwc_empty(N, C, H, W, x.dtype, x.device)
    dres = nhwc_empty(N, C, H, W, x.dtype, x.device) if want_dres else None
    if partial is not None:
        buf, ns = partial
        coeff = torch.empty(3 * C, dtype=torch.float32, device=x.device)
        call('mi355_bn_bwd_partials', ptr(dy), ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dx),
             ptr(dres), ptr(dgamma), ptr(dbeta), int(accumulate), rows, C, int(relu), dtype_code(x.dtype), ptr(buf), int(ns),
             ptr(coeff), ptr(relu_mask), ptr(q8_out), ptr(q8_state), stream_ptr())
        return dx, dres
    ws = workspace(load().mi355_bn_workspace(rows, C), x.device)
    call('mi355_bn_bwd', ptr(dy), ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dx), ptr(dres),
         ptr(dgamma), ptr(dbeta), int(accumulate), rows, C, int(relu), dtype_code(x.dtype), ptr(ws), ws.numel(),
         ptr(relu_mask), ptr(q8_out), ptr(q8_state), stream_ptr())
    return dx, dres


# ---------------------------------------------------------------- max pool
def bn_resident_timeouts():
    """Grid-barrier spins of the one-launch BatchNorm backward that gave up since the library was loaded or the last reset
    (0 = healthy).  Every such launch has written NaN into the gradients it produced.  Synchronises the device."""
    n = ctypes.c_uint(0)
    call('mi355_bn_resident_timeouts', ctypes.byref(n))
    return n.value


def bn_resident_reset():
    call('mi355_bn_resident_reset')


def bn_resident_set_spin_limit(limit):
    """Test hook: poll iterations before a block of the one-launch BatchNorm backward gives up (0 = default)."""
    call('mi355_bn_resident_set_spin_limit', int(limit))


def bn_resident_check(where=''):
    """Raise when a one-launch BatchNorm backward could not get all its blocks onto the chip since the last check: its gradients
    are NaN (csrc/bn.hip bn_res_grid_barrier).  The one-launch form is switched off for the rest of the process and the counters
    are cleared, so a caller that catches the error can redo the step on the three-launch path.  Synchronises the device: call it
    where the host waits anyway (train1.py: once per epoch; bench.py / smoke: at the end)."""
    n = bn_resident_timeouts()
    if n:
        load().mi355_bn_set_resident(0)
        bn_resident_reset()
        raise Mi355Error('%d grid-barrier give-up(s) in the one-launch BatchNorm backward%s: another kernel or process held CUs while '
                         'it ran, the gradients of those launches are NaN.  The one-launch form is now off for this process '
                         '(MI355_BN_RESIDENT=0 makes that the default); redo the affected steps.' % (n, (' (' + where + ')') if where else ''))


def maxpool_fwd(x):
    N, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = nhwc_empty(N, C, Ho, Wo, x.dtype, x.device)
    arg = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    call('mi355_maxpool_fwd', ptr(x), ptr(y), ptr(arg), N, H, W, C, dtype_code(x.dtype), stream_ptr())
    return y, arg


def maxpool_bwd(dy, arg, in_shape):
    N, C, H, W = in_shape
    dx = nhwc_empty(N, C, H, W, dy.dtype, dy.device)
    call('mi355_maxpool_bwd', ptr(dy), ptr(arg), ptr(dx), N, H, W, C, dtype_code(dy.dtype), stream_ptr())
    return dx


# ---------------------------------------------------------------- 21-channel pointwise convs
def conv1x1_heatmap(x, w_packed, bias, K):
    """MFMA form of the C -> K heat-map conv: x channels_last [N,C,H,W], w_packed [K][C] in x.dtype -> [N,K,H,W] fp32."""
    N, C, H, W = x.shape
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    call('mi355_conv1x1_heatmap', ptr(x), ptr(w_packed), ptr(bias), ptr(y), N, H * W, C, K, dtype_code(x.dtype), stream_ptr())
    return y


def pw_c2k(x, w, bias, K, w_transposed=False):
    """x channels_last [N,C,H,W] -> heat-map [N,K,H,W] fp32 contiguous."""
    N, C, H, W = x.shape
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    call('mi355_pw_c2k', ptr(x), ptr(w), ptr(bias), ptr(y), N, H * W, C, K, int(w_transposed), dtype_code(x.dtype),
         stream_ptr())
    return y


def pw_k2c(y, w, bias, C, dtype, residual=None, scale_dev=None, w_transposed=False):
    """heat-map [N,K,H,W] fp32 -> channels_last [N,C,H,W] `dtype`."""
    N, K, H, W = y.shape
    out = nhwc_empty(N, C, H, W, dtype, y.device)
    call('mi355_pw_k2c', ptr(y), ptr(w), ptr(bias), ptr(residual), ptr(scale_dev), ptr(out), N, H * W, C, K,
         int(w_transposed), dtype_code(dtype), stream_ptr())
    return out


def pw_k2c_stats(y, w, bias, C, dtype, residual=None):
    """pw_k2c + BatchNorm statistics partials of its output.  Returns (out, (partial, nslices))."""
    N, K, H, W = y.shape
    out = nhwc_empty(N, C, H, W, dtype, y.device)
    ns_max = N * ((H * W + 63) // 64)
    partial = torch.empty(ns_max * C * 3, dtype=torch.float32, device=y.device)
    ns = ctypes.c_int(0)
    call('mi355_pw_k2c_stats', ptr(y), ptr(w), ptr(bias), ptr(residual), None, ptr(out), N, H * W, C, K, 0,
         dtype_code(dtype), ptr(partial), partial.numel() * 4, ctypes.byref(ns), stream_ptr())
    return out, (partial, ns.value)


def pw_wgrad(x, y, dw, kc_layout, accumulate):
    N, C, H, W = x.shape
    K = y.shape[1]
    ws = workspace(load().mi355_pw_wgrad_workspace(N, H * W, C, K), x.device)
    call('mi355_pw_wgrad', ptr(x), ptr(y), ptr(dw), int(kc_layout), int(accumulate), N, H * W, C, K,
         dtype_code(x.dtype), ptr(ws), ws.numel(), stream_ptr())


def hm_rowsum(y, out, accumulate):
    N, K, H, W = y.shape
    ws = workspace(N * K * 4, y.device)
    call('mi355_hm_rowsum', ptr(y), ptr(out), int(accumulate), N, K, H * W, ptr(ws), ws.numel(), stream_ptr())


# ---------------------------------------------------------------- heat-map decode / losses
def _hm(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    _chk_dev(t)
    return t


def argmax2d(hm):
    """(idx int32 [B,K], xy fp32 [B,K,2], maxval fp32 [B,K,1]) with numpy's first-max tie rule."""
    hm = _hm(hm)
    B, K, H, W = hm.shape
    idx = torch.empty((B, K), dtype=torch.int32, device=hm.device)
    xy = torch.empty((B, K, 2), dtype=torch.float32, device=hm.device)
    mv = torch.empty((B, K, 1), dtype=torch.float32, device=hm.device)
    call('mi355_argmax2d', ptr(hm), ptr(idx), ptr(xy), ptr(mv), B * K, H, W, stream_ptr())
    return idx, xy, mv


def softargmax(hm, beta=100.0, out_scale=4.0):
    hm = _hm(hm)
    B, K, H, W = hm.shape
    uv = torch.empty((B, K, 2), dtype=torch.float32, device=hm.device)
    call('mi355_softargmax', ptr(hm), ptr(uv), B * K, H, W, float(beta), float(out_scale), stream_ptr())
    return uv


def kl_heatmap(pred, target, weight, eps, want_grad, coeff=1.0):
    """Returns (loss_rows [B,K], unit_grad [B,K,H,W] or None); unit_grad = d(coeff * mean over B*K)/d pred."""
    pred, target = _hm(pred), _hm(target)
    B, K, H, W = pred.shape
    if tuple(target.shape) != (B, K, H, W):
        raise Mi355Error('kl_heatmap: pred %s vs target %s' % (tuple(pred.shape), tuple(target.shape)))
    rows = torch.empty((B, K), dtype=torch.float32, device=pred.device)
    g = torch.empty_like(pred) if want_grad else None
    if weight is not None:
        weight = weight.reshape(B, K).float().contiguous()
    call('mi355_kl_heatmap', ptr(pred), ptr(target), ptr(weight), float(eps), ptr(rows), ptr(g), B * K, H * W,
         float(coeff) / (B * K), stream_ptr())
    return rows, g


def reduce_sum(v, scale=1.0):
    v = v.contiguous()
    out = torch.empty((), dtype=torch.float32, device=v.device)
    call('mi355_reduce_sum', ptr(v), ptr(out), v.numel(), float(scale), stream_ptr())
    return out


def scale_by_dev(t, g_dev):
    out = torch.empty_like(t)
    call('mi355_scale_by_dev', ptr(t), ptr(g_dev), ptr(out), t.numel(), stream_ptr())
    return out


def scale_feature(t, g_dev):
    """t * (*g_dev) for a dense fp32 / bf16 device tensor of any layout (same strides out)."""
    if not t.is_cuda:
        raise Mi355Error('mi355 ops need CUDA/HIP tensors; there is no CPU fallback')
    if not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        t = t.contiguous()
    per = 8 if t.dtype == torch.bfloat16 else 4
    if t.numel() % per:
        raise Mi355Error('scale_feature needs a multiple of %d elements, got %d' % (per, t.numel()))
    out = torch.empty_like(t)
    call('mi355_scale_feature', ptr(t), ptr(g_dev), ptr(out), t.numel(), dtype_code(t.dtype), stream_ptr())
    return out


def pseudo_label(xy, patch, radius, div, S, kind, extra=None, normalise=False, want_gt=True, want_gf=True):
    B, K, _ = xy.shape
    dev = xy.device
    gt = torch.empty((B, K, S, S), dtype=torch.float32, device=dev) if want_gt else None
    gf = torch.empty((B, K, S, S), dtype=torch.float32, device=dev) if want_gf else None
    if extra is not None:
        extra = _hm(extra)
        if tuple(extra.shape) != (B, K, S, S):
            raise Mi355Error('pseudo_label: extra has shape %s, expected %s' % (tuple(extra.shape), (B, K, S, S)))
    call('mi355_pseudo_label', ptr(xy), ptr(patch), int(radius), int(div), int(S), int(kind), ptr(extra),
         int(normalise), ptr(gt), ptr(gf), B, K, stream_ptr())
    return gt, gf


def bilinear_up(x, size, alpha=1.0, out=None):
    """alpha * nn.Upsample(size, mode='bilinear')(x) (+ out when given)."""
    x = _hm(x)
    B, K, h, w = x.shape
    acc = out is not None
    if out is None:
        out = torch.empty((B, K, size, size), dtype=torch.float32, device=x.device)
    call('mi355_bilinear_up', ptr(x), ptr(out), B * K, h, w, size, size, float(alpha), int(acc), stream_ptr())
    return out


def pck_dists(pred_xy, tgt_xy, norm_x, norm_y):
    rows = pred_xy.shape[0] * pred_xy.shape[1]
    d = torch.empty(pred_xy.shape[:2], dtype=torch.float32, device=pred_xy.device)
    call('mi355_pck_dists', ptr(pred_xy.contiguous()), ptr(tgt_xy.contiguous()), ptr(d), rows, float(norm_x),
         float(norm_y), stream_ptr())
    return d


# ---------------------------------------------------------------- optimiser
def sgd_nesterov(p, g, buf, lr_dev, momentum, wd, nesterov, p_lowp=None):
    call('mi355_sgd_nesterov', ptr(p), ptr(g), ptr(buf), p.numel(), ptr(lr_dev), float(momentum), float(wd),
         int(nesterov), ptr(p_lowp), stream_ptr())


def cast_f32(src, dst):
    call('mi355_cast_f32', ptr(src), ptr(dst), src.numel(), dtype_code(dst.dtype), stream_ptr())


# ---------------------------------------------------------------- kernel timer (bench.py roofline)
def spin_us(us):
    call('mi355_spin_us', int(us), stream_ptr())


def prof_event_overhead_us(n=256):
    us = ctypes.c_double(0)
    call('mi355_prof_event_overhead_us', int(n), stream_ptr(), ctypes.byref(us))
    return us.value


def prof_read_split(flop_per_byte):
    out = (ctypes.c_double * 8)()
    call('mi355_prof_read_split', float(flop_per_byte), out)
    return list(out)


def prof_enable(on):
    call('mi355_prof_enable', int(on))


def prof_reset():
    call('mi355_prof_reset')


def prof_launches():
    """Every launch logged since the last reset, in launch order: dicts with family, us (event-timed), flops, bytes, label."""
    n = ctypes.c_long()
    call('mi355_prof_launch_count', ctypes.byref(n))
    out = []
    fam, us, fl, by = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    buf = ctypes.create_string_buffer(192)
    for i in range(n.value):
        call('mi355_prof_read_launch', i, ctypes.byref(fam), ctypes.byref(us), ctypes.byref(fl), ctypes.byref(by), buf, 192)
        out.append({'family': fam.value, 'us': us.value, 'flops': fl.value, 'bytes': by.value, 'label': buf.value.decode()})
    return out


def prof_read():
    ms, n, fl, by = ctypes.c_double(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double()
    call('mi355_prof_read', ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
    return ms.value, n.value, fl.value, by.value
