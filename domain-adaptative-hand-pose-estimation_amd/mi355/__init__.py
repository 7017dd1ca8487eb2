"""ctypes binding of libmi355pose.so (include/mi355pose.h).

The product path has NO CPU / PyTorch fallback: every op raises if the HIP library is missing or a
call fails.  PyTorch is used for device memory, streams and autograd bookkeeping only.
"""
import ctypes
import os
import threading

import torch

F32, BF16, FP8 = 0, 1, 2
_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_LIB: load another build of the same library instead (A/B runs of compiler flags: build.py --variant; the bisection
# builds of profiles/dropped_addend_repro.py) -- an experiment switch, unset in normal use
LIB_PATH = os.environ.get('MI355_LIB') or os.path.join(os.path.dirname(_HERE), 'libmi355pose.so')

_lib = None
_lock = threading.Lock()


class Mi355Error(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ('N', 'Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'kh', 'kw', 'stride', 'pad', 'dtype')]


class WgradItem(ctypes.Structure):
    """mi355_wgrad_item: one problem of mi355_conv_wgrad_grouped."""
    _fields_ = [('d', ConvDesc), ('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p),
                ('accumulate', ctypes.c_int), ('pad_', ctypes.c_int)]


class BnBwdSrc(ctypes.Structure):
    """mi355_bn_bwd_src: saved forward state of the BatchNorm whose dy a GEMM epilogue reduces."""
    _fields_ = [('x', ctypes.c_void_p), ('y', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p),
                ('save_mean', ctypes.c_void_p), ('save_invstd', ctypes.c_void_p), ('relu', ctypes.c_int)]


# name -> (restype, argtypes); mirrors include/mi355pose.h one to one
_P, _I, _L, _F, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t
_D = ctypes.POINTER(ConvDesc)
SIGNATURES = {
    'mi355_version': (_I, []),
    'mi355_last_error': (ctypes.c_char_p, []),
    'mi355_conv_fwd': (_I, [_D, _P, _P, _P, _P, _P, _P]),
    'mi355_conv_dgrad': (_I, [_D, _P, _P, _P, _P, _I, _P, _P]),
    'mi355_conv_stats_bytes': (_Z, [_L, _I]),
    'mi355_conv_fwd_stats': (_I, [_D, _P, _P, _P, _P, _P, _Z, _P, _P]),
    'mi355_conv_dgrad_stats': (_I, [_D, _P, _P, _P, _P, _Z, _P, _P]),
    'mi355_conv_fwd_cat': (_I, [_D, _P, _P, _P, _P, _P, _P, _I, _P, _P, _Z, _P, _P]),
    'mi355_conv_dgrad_bnbwd': (_I, [_D, _P, _P, _P, _I, _P, _P, _P, _Z, _P, _P]),
    'mi355_conv_fwd_bnbwd': (_I, [_D, _P, _P, _P, _P, _P, _Z, _P, _P]),
    'mi355_fp8_quantize': (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    'mi355_fp8_update_scale': (_I, [_P, _I, _I, _I, _I, _P]),
    'mi355_pack_weights_fp8': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'mi355_pack_weights_fp8_batched': (_I, [_P, _I, _I, _P]),
    'mi355_conv_fwd_fp8': (_I, [_D, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _P]),
    'mi355_conv_dgrad_fp8': (_I, [_D, _P, _I, _P, _P, _P, _P, _I, _P, _P, _Z, _P, _P]),
    'mi355_conv_wgrad_workspace': (_Z, [_D]),
    'mi355_conv_wgrad_fp8_workspace': (_Z, [_D]),
    'mi355_conv_wgrad_fp8': (_I, [_D, _P, _I, _P, _I, _P, _P, _P, _I, _P, _Z, _P]),
    'mi355_conv_wgrad': (_I, [_D, _P, _P, _P, _I, _P, _Z, _P]),
    'mi355_conv_wgrad_grouped_workspace': (_Z, [_P, _I]),
    'mi355_conv_wgrad_grouped': (_I, [_P, _I, _P, _Z, _P]),
    'mi355_pack_weights': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mi355_pack_weights_batched': (_I, [_P, _I, _I, _I, _P]),
    'mi355_colsum_workspace': (_Z, [_L, _I]),
    'mi355_colsum': (_I, [_P, _P, _L, _I, _I, _I, _P, _Z, _P]),
    'mi355_bn_workspace': (_Z, [_L, _I]),
    'mi355_bn_train_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _I, _I, _I, _P, _Z, _P, _P, _P, _P]),
    'mi355_bn_train_fwd_partials': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    'mi355_bn_eval_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _I, _P]),
    'mi355_bn_bwd_partials': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    'mi355_bn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P, _Z, _P, _P, _P, _P]),
    'mi355_bn_resident_timeouts': (_I, [_P]),
    'mi355_bn_resident_reset': (_I, []),
    'mi355_bn_resident_set_spin_limit': (_I, [ctypes.c_uint]),
    'mi355_bn_set_resident': (_I, [_I]),
    'mi355_set_pgemm': (_I, [_I]),
    'mi355_set_fp8_kw3': (_L, [_L]),
    'mi355_apply_relu_mask': (_I, [_P, _P, _L, _I, _I, _P]),
    'mi355_conv_dgrad_masked_acc': (_I, [_P, _P, _P, _P, _P, _P, _P]),
    'mi355_conv_fwd_act': (_I, [_P, _P, _P, _P, _P, _I, _P, _P]),
    'mi355_conv_dgrad_act': (_I, [_P, _P, _P, _P, _I, _P, _P]),
    'mi355_bn_relu_maxpool_fwd_partials': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _I, _P, _I, _P, _P]),
    'mi355_maxpool_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mi355_maxpool_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mi355_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mi355_nchw_to_s2d': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'mi355_stem_s2d_pack': (_I, [_P, _P, _I, _I, _P]),
    'mi355_stem_s2d_unpack_grad': (_I, [_P, _P, _I, _I, _P]),
    'mi355_nhwc_to_nchw': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'mi355_conv1x1_heatmap': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mi355_pw_c2k': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mi355_pw_k2c': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mi355_pw_k2c_stats': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P]),
    'mi355_pw_wgrad_workspace': (_Z, [_I, _I, _I, _I]),
    'mi355_pw_wgrad': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    'mi355_hm_rowsum': (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    'mi355_argmax2d': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'mi355_softargmax': (_I, [_P, _P, _I, _I, _I, _F, _F, _P]),
    'mi355_kl_heatmap': (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _F, _P]),
    'mi355_reduce_sum': (_I, [_P, _P, _I, _F, _P]),
    'mi355_scale_by_dev': (_I, [_P, _P, _P, _L, _P]),
    'mi355_scale_feature': (_I, [_P, _P, _P, _L, _I, _P]),
    'mi355_pseudo_label': (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I, _P]),
    'mi355_bilinear_up': (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    'mi355_pck_dists': (_I, [_P, _P, _P, _I, _F, _F, _P]),
    'mi355_sgd_nesterov': (_I, [_P, _P, _P, _L, _P, _F, _F, _I, _P, _P]),
    'mi355_cast_f32': (_I, [_P, _P, _L, _I, _P]),
    'mi355_prof_enable': (_I, [_I]),
    'mi355_spin_us': (_I, [_L, _P]),
    'mi355_prof_read_split': (_I, [ctypes.c_double, ctypes.POINTER(ctypes.c_double)]),
    'mi355_prof_event_overhead_us': (_I, [_I, _P, ctypes.POINTER(ctypes.c_double)]),
    'mi355_prof_reset': (_I, []),
    'mi355_prof_launch_count': (_I, [ctypes.POINTER(ctypes.c_long)]),
    'mi355_prof_read_launch': (_I, [_L, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, _I]),
    'mi355_prof_read': (_I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long),
                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}


def load(path=None):
    """Load the shared library (once).  Raises Mi355Error when it is missing: there is no fallback."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise Mi355Error(
                'libmi355pose.so not found at %s: build it with '
                '`python domain-adaptative-hand-pose-estimation_amd/build.py` (hipcc, gfx950). '
                'The MI355X path has no CPU fallback.' % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


_fn_cache = {}


def call(name, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise Mi355Error('%s failed (%d): %s' % (name, rc, load().mi355_last_error().decode()))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """HIP stream of the calling thread's current torch stream, as an integer (one C call: torch.cuda.current_stream() builds a
    Python Stream object per launch, ~10 us of the ~25 us a launch costs on the host)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


# ---------------------------------------------------------------- compute dtype (activations + packed weights)
_compute_dtype = torch.bfloat16


_fp8_convs = False


def set_compute_dtype(dt):
    """'bf16' (default, throughput path), 'f32' (exact-fp32 MFMA parity path) or 'fp8': bf16 storage and kernels
    everywhere, except that the forward and input-gradient GEMMs of the K-heavy convolutions (3x3 / 4x4, channel counts
    that are multiples of 128) run on fp8 operands (e4m3 activations and weights, e5m2 gradients, per-tensor delayed
    scaling, fp32 accumulate); weight gradients stay bf16."""
    global _compute_dtype, _fp8_convs
    if dt in ('bf16', torch.bfloat16):
        _compute_dtype, _fp8_convs = torch.bfloat16, False
    elif dt in ('f32', 'fp32', torch.float32):
        _compute_dtype, _fp8_convs = torch.float32, False
    elif dt in ('fp8', 'f8'):
        _compute_dtype, _fp8_convs = torch.bfloat16, True
    else:
        raise ValueError('compute dtype must be bf16, f32 or fp8, got %r' % (dt,))


def graph_capture_mode():
    """`capture_error_mode` for torch.cuda.graph on this process.  With an RCCL ('nccl') process group initialised, the group's
    watchdog thread polls the events of earlier collectives with hipEventQuery; a capture in the default 'global' mode forbids that
    call to EVERY thread, the watchdog dies with hipErrorStreamCaptureUnsupported and std::terminate takes the process with it (a
    race, seen once in tests/rccl_single_rank.py).  'thread_local' restricts the check to the capturing thread; the device is
    drained and the watchdog given a moment to retire what is pending.  Without such a group: the default."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        import time
        torch.cuda.synchronize()
        time.sleep(0.2)
        return 'thread_local'
    return 'global'


def compute_dtype():
    """Storage / kernel dtype of activations (bf16 in 'fp8' mode as well)."""
    return _compute_dtype


def fp8_convs():
    return _fp8_convs


# ---------------------------------------------------------------- fp8 scaling states (delayed scaling)
# Every fp8 operand stream (one per conv input, one per conv output gradient) owns a 4-float device record
# {scale, descale, amax bits, pad} carved out of one pool per format, so that ONE launch per format refreshes every
# scale from the amax values recorded since the last refresh (`fp8_tick`, called once per optimizer step).
_FP8_POOL_SLOTS = 2048
_fp8_pools = {}


def fp8_alloc_state(device, fmt):
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(fmt))
    pool = _fp8_pools.get(key)
    if pool is None:
        pool = _fp8_pools[key] = dict(buf=torch.zeros(_FP8_POOL_SLOTS * 4, dtype=torch.float32, device=device), used=0, fmt=int(fmt))
    if pool['used'] >= _FP8_POOL_SLOTS:
        raise Mi355Error('fp8 scaling-state pool exhausted')
    i = pool['used']
    pool['used'] += 1
    return pool['buf'][4 * i: 4 * i + 4]


def fp8_tick():
    """Derive every fp8 scale from the amax recorded since the previous tick (delayed scaling); one launch per format."""
    for pool in _fp8_pools.values():
        if pool['used']:
            call('mi355_fp8_update_scale', ptr(pool['buf']), pool['used'], 4, pool['fmt'], 0, stream_ptr())


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise Mi355Error('unsupported dtype %s' % dt)


# ---------------------------------------------------------------- BatchNorm running-stat update multiplicity
bn_stat_updates = 1


class bn_updates:
    """with bn_updates(2): forwards inside apply the running-stat momentum update twice (one forward standing for two
    identical forwards of the reference loop)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        global bn_stat_updates
        self.prev, bn_stat_updates = bn_stat_updates, self.n

    def __exit__(self, *exc):
        global bn_stat_updates
        bn_stat_updates = self.prev


# ---------------------------------------------------------------- shared scratch (stream-ordered reuse)
_ws = {}


def workspace(nbytes, device, tag='main'):
    """One grow-only scratch buffer per (device, stream role); every op of one role is enqueued on one stream,
    so reuse is ordered.  Grows outside graph capture only (warm up before capturing)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise Mi355Error('workspace would grow (%d bytes) during graph capture: warm up first' % nbytes)
        size = max(int(nbytes), 64 << 20)
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


import os as _os

# ---------------------------------------------------------------- grouped weight gradients
# Inside `with grouped_wgrads():` (the training step wraps each forward+backward in it) the conv layers do not launch their
# weight gradients where autograd reaches them; the problems are collected and handed to mi355_conv_wgrad_grouped in batches
# -- at the stage boundaries of the backward (DAStep._on_stage_grad) and when the block ends -- so that the many small
# layers of a ResNet stage share one launch.  Outside such a block every weight gradient is launched immediately
# (a caller may read param.grad right after backward()).
# (Weight gradients on a side stream -- per layer, collected per stage, or the grouped launch -- were measured slower in every
#  form, rounds 1-3: each fork / join inside the captured graphs costs more than the overlap returns; DESIGN.md section 7.  That
#  machinery is gone; weight gradients run on the stream of the backward pass.)
GROUP_WGRAD = _os.environ.get('MI355_WGRAD_GROUP', '1') == '1'
_group_depth = 0
_group_items = []


class grouped_wgrads:
    def __enter__(self):
        global _group_depth
        _group_depth += 1

    def __exit__(self, *exc):
        global _group_depth
        _group_depth -= 1
        if _group_depth == 0:
            if exc[0] is None:
                flush_grouped_wgrads()
            else:
                del _group_items[:]


def grouping_wgrads():
    return GROUP_WGRAD and _group_depth > 0


def group_wgrad(desc, x, dy, dw, accumulate):
    _group_items.append((desc, x, dy, dw, bool(accumulate)))


def flush_grouped_wgrads():
    if not _group_items:
        return
    from . import ops
    items = list(_group_items)
    del _group_items[:]
    ops.conv_wgrad_grouped(items)


# ---------------------------------------------------------------- unit gradient of a total loss
# `total.backward(unit_grad(total))` instead of `total.backward()`: autograd hands this very tensor (sums pass their incoming
# gradient on untouched) to the loss Functions, which recognise it by address and skip the multiplication by 1.
_unit = {}


def unit_grad(like):
    key = (like.device.type, like.device.index)
    u = _unit.get(key)
    if u is None:
        u = _unit[key] = torch.ones((), dtype=torch.float32, device=like.device)
    return u


def is_unit_grad(g):
    u = _unit.get((g.device.type, g.device.index))
    return u is not None and g.dim() == 0 and g.dtype == torch.float32 and g.data_ptr() == u.data_ptr()


def join_side():
    """Everything a backward pass has deferred is enqueued: the weight gradients still waiting for their grouped launch."""
    flush_grouped_wgrads()
