"""Heat-map -> feature 1x1 conv (K=21 -> C=256; the input gradient of the heads' last conv and the forward of `heatmap_conv`):
the fp32 VALU kernel against the same product on the MFMA gather kernel (heat-maps re-laid as NHWC bf16 with K padded to 32).
usage: python profiles/pw_k2c_bench.py"""
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops

dev = torch.device('cuda:0'); mi355.load(); dt = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for S in (64, 32, 16):
    N, K, C = 64, 21, 256
    hm = torch.randn(N, K, S, S, device=dev)
    w = torch.randn(C, K, device=dev) * 0.1
    res = ops.nhwc_empty(N, C, S, S, dt, dev).normal_()
    t0 = timeit(lambda: ops.pw_k2c(hm, w, None, C, dt))
    t1 = timeit(lambda: ops.pw_k2c(hm, w, None, C, dt, residual=res))
    t2 = timeit(lambda: ops.pw_k2c_stats(hm, w, None, C, dt, residual=res))
    wp = torch.zeros(C, 32, device=dev); wp[:, :K] = w
    wf = wp.to(dt).contiguous().view(-1)
    desc = ops.make_desc(N, S, S, 32, C, 1, 1, 1, 0, dt)
    t3 = timeit(lambda: ops.to_nhwc(hm, dt, 32))
    h32 = ops.to_nhwc(hm, dt, 32)
    t4 = timeit(lambda: ops.conv_fwd(desc, h32, wf))
    t5 = timeit(lambda: ops.conv_fwd(desc, h32, wf, residual=res))
    t6 = timeit(lambda: ops.conv_fwd_stats(desc, h32, wf))
    a = ops.pw_k2c(hm, w, None, C, dt).float(); b = ops.conv_fwd(desc, h32, wf).float()
    err = float((a - b).norm() / a.norm())
    mb = N * S * S * C * 2 / 1e6
    print('%dx%d: VALU %.1f us (+residual %.1f, +stats %.1f) | relayout %.1f + MFMA %.1f (+residual %.1f, stats %.1f) us | '
          'out %.0f MB | rel diff %.1e' % (S, S, t0, t1, t2, t3, t4, t5, t6, mb, err))
