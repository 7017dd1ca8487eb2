"""Forward / dgrad time of the mid-size conv shapes under the gather-kernel tile switch MI355_TILE (one value per process).
usage: MI355_TILE=<n> python profiles/tile_sweep.py"""
import os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops
dev = torch.device('cuda:0'); mi355.load(); dt = torch.bfloat16; B = 64


def timeit(fn, n=30):
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


out = []
for (H, Ci, Co, k, s, p) in [(16, 256, 256, 3, 1, 1), (8, 512, 512, 3, 1, 1), (16, 1024, 256, 1, 1, 0), (16, 256, 1024, 1, 1, 0),
                             (32, 128, 128, 3, 1, 1), (32, 512, 128, 1, 1, 0), (32, 128, 512, 1, 1, 0), (8, 2048, 512, 1, 1, 0), (8, 512, 2048, 1, 1, 0),
                             (32, 256, 256, 3, 2, 1)]:
    d = ops.make_desc(B, H, H, Ci, Co, k, k, s, p, dt)
    x = ops.nhwc_empty(B, Ci, H, H, dt, dev).normal_(); dy = ops.nhwc_empty(B, Co, d.Ho, d.Wo, dt, dev).normal_()
    w = (torch.randn(Co * k * k * Ci, device=dev) * 0.02)
    wf, wt = ops.pack_weights(w, Co, k * k, Ci, Ci, dt)
    out.append('%dx%d s%d %d->%d @%d: fwd %.1f dgrad %.1f' % (k, k, s, Ci, Co, H, timeit(lambda: ops.conv_fwd(d, x, wf)), timeit(lambda: ops.conv_dgrad(d, dy, wt))))
print('TILE', os.environ.get('MI355_TILE', 'auto'), ' | '.join(out))
