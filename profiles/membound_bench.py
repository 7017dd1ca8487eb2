"""Achieved HBM bandwidth of the memory-bound stages (BatchNorm forward / backward, arg-max and soft-arg-max decode, KL loss)
in isolation at the tensor sizes of the ResNet-50 / 256x256 / B=64 iteration.  Algorithmic bytes: BN forward 3 passes of the
tensor (2 reads + 1 write; 2 when the conv fused the statistics), BN backward 5 passes, decode / loss: every input element once.
Graph-timed (50 launches per replay) so that launch gaps do not count.   usage: python profiles/membound_bench.py"""
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops

dev = torch.device('cuda:0'); mi355.load(); dt = torch.bfloat16


import os
EAGER = os.environ.get('MEMBOUND_EAGER') == '1'      # under rocprofv3 --pmc: plain launches, no graphs, counters per kernel


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if EAGER:
        return float('nan')
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


print('%-34s %9s %9s %9s' % ('stage (tensor)', 'MB moved', 'us', 'TB/s'))
for (N, C, H) in [(64, 256, 64), (64, 64, 64), (64, 512, 32), (64, 128, 32), (64, 1024, 16), (64, 256, 16), (64, 2048, 8), (64, 512, 8)]:
    x = ops.nhwc_empty(N, C, H, H, dt, dev).normal_(); dy = ops.nhwc_empty(N, C, H, H, dt, dev).normal_()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    y, mean, invstd = ops.bn_train_fwd(x, None, g, b, rm, rv, nbt, 1e-5, 0.1, True)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    mb = N * C * H * H * 2 / 1e6
    t = timeit(lambda: ops.bn_train_fwd(x, None, g, b, rm, rv, nbt, 1e-5, 0.1, True))
    print('%-34s %9.0f %9.1f %9.2f' % ('BN fwd  %dx%dx%dx%d bf16' % (N, C, H, H), 3 * mb, t, 3 * mb / t))
    t = timeit(lambda: ops.bn_bwd(dy, x, None, g, mean, invstd, dg, db, False, True, False, beta=b))
    print('%-34s %9.0f %9.1f %9.2f' % ('BN bwd  %dx%dx%dx%d bf16' % (N, C, H, H), 5 * mb, t, 5 * mb / t))
for S in (64, 32, 16):
    hm = torch.randn(64, 21, S, S, device=dev); tg = torch.rand(64, 21, S, S, device=dev); w = torch.ones(64, 21, 1, device=dev)
    mb = hm.numel() * 4 / 1e6
    t = timeit(lambda: ops.argmax2d(hm))
    print('%-34s %9.1f %9.1f %9.2f' % ('arg-max decode 64x21x%dx%d fp32' % (S, S), mb, t, mb / t))
    t = timeit(lambda: ops.softargmax(hm))
    print('%-34s %9.1f %9.1f %9.2f' % ('soft-arg-max   64x21x%dx%d fp32' % (S, S), mb, t, mb / t))
    t = timeit(lambda: ops.kl_heatmap(hm, tg, w, 1e-7, True))
    print('%-34s %9.1f %9.1f %9.2f' % ('KL loss + grad 64x21x%dx%d fp32' % (S, S), 3 * mb, t, 3 * mb / t))
