"""fp8 vs bf16 conv GEMMs on the K-heavy layer shapes (graph-timed in isolation): forward, input gradient, weight gradient.
usage: python profiles/fp8_layer_bench.py [B]"""
import sys
import torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops

dev = torch.device('cuda:0'); mi355.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('%-40s %9s %9s %9s %9s %9s' % ('layer', 'bf16 us', 'fp8 us', 'TF/s bf16', 'TF/s fp8', 'quant us'))
for (H, Ci, Co, k, s, p) in [(64, 256, 256, 3, 1, 1), (64, 256, 256, 3, 2, 1), (32, 256, 256, 3, 1, 1), (32, 128, 128, 3, 1, 1),
                             (16, 256, 256, 3, 1, 1), (8, 512, 512, 3, 1, 1), (64, 256, 256, 4, 2, 1), (64, 256, 256, 1, 1, 0),
                             (16, 1024, 256, 1, 1, 0)]:
    dt = torch.bfloat16
    d16 = ops.make_desc(B, H, H, Ci, Co, k, k, s, p, dt)
    d8 = ops.make_desc_fp8(B, H, H, Ci, Co, k, k, s, p)
    x = ops.nhwc_empty(B, Ci, H, H, dt, dev).normal_(); dy = ops.nhwc_empty(B, Co, d16.Ho, d16.Wo, dt, dev).normal_()
    w = torch.randn(Co * k * k * Ci, device=dev) * 0.02
    wf, wt = ops.pack_weights(w, Co, k * k, Ci, Ci, dt)
    sx, sw, sd = ops.fp8_state(dev), ops.fp8_state(dev), ops.fp8_state(dev)
    wf8, wt8 = ops.pack_weights_fp8(w, Co, k * k, Ci, sw)
    x8 = ops.fp8_quantize(x, sx, ops.E4M3, jit=True); dy8 = ops.fp8_quantize(dy, sd, ops.E5M2, jit=True)
    fl = 2.0 * B * d16.Ho * d16.Wo * Co * k * k * Ci
    kinds = ('fwd', 'dgrad', 'wgrad') if (k == 3 and s == 1) else ('fwd', 'dgrad')
    dw = torch.zeros(Co * k * k * Ci, device=dev)
    for kind in kinds:
        if kind == 'fwd':
            t16 = timeit(lambda: ops.conv_fwd(d16, x, wf)); t8 = timeit(lambda: ops.conv_fwd_fp8(d8, x8, sx, wf8, sw))
            tq = timeit(lambda: ops.fp8_quantize(x, sx, ops.E4M3))
        elif kind == 'wgrad':     # 3x3 / stride 1: weight gradient from the two fp8 copies the other GEMMs already use (no extra pass)
            t16 = timeit(lambda: ops.conv_wgrad(d16, x, dy, dw, False)); t8 = timeit(lambda: ops.conv_wgrad_fp8(d8, x8, sx, dy8, sd, dw, False))
            tq = 0.0
        else:
            t16 = timeit(lambda: ops.conv_dgrad(d16, dy, wt)); t8 = timeit(lambda: ops.conv_dgrad_fp8(d8, dy8, sd, wt8, sw))
            tq = timeit(lambda: ops.fp8_quantize(dy, sd, ops.E5M2))
        print('%-40s %9.1f %9.1f %9.0f %9.0f %9.1f' % ('%s %dx%d s%d %d->%d @%d' % (kind, k, k, s, Ci, Co, H), t16, t8, fl / t16 / 1e6, fl / t8 / 1e6, tq))
