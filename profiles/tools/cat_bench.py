import sys, os, torch
R = os.getcwd(); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops
mi355.load()
dev = torch.device('cuda:0')
dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, Ci, H, W, Co, k, s, p) in [(64, 256, 64, 64, 256, 1, 1, 0), (64, 256, 64, 64, 256, 3, 2, 1)]:
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, dt)
    x = ops.nhwc_empty(N, Ci, H, W, dt, dev).normal_()
    wm = (torch.randn(Co, k, k, Ci, device=dev) * 0.02)
    wf, wt = ops.pack_weights(wm, Co, k * k, Ci, Ci, dt)
    hm32 = ops.nhwc_empty(N, 32, desc.Ho, desc.Wo, dt, dev).normal_()
    w2 = torch.randn(Co, 32, device=dev).to(dt)
    b = torch.zeros(Co, device=dev)
    hm = torch.randn(N, 21, desc.Ho, desc.Wo, device=dev)
    wh = torch.randn(Co, 21, device=dev)
    t_plain = timeit(lambda: ops.conv_fwd(desc, x, wf, b))
    t_stats = timeit(lambda: ops.conv_fwd_stats(desc, x, wf, b))
    t_cat = timeit(lambda: ops.conv_fwd_cat(desc, x, wf, b, hm32, w2, b))
    t_cats = timeit(lambda: ops.conv_fwd_cat(desc, x, wf, b, hm32, w2, b, want_stats=True))
    y = ops.conv_fwd(desc, x, wf, b)
    t_k2c = timeit(lambda: ops.pw_k2c_stats(hm, wh, b, Co, dt, residual=y))
    t_cvt = timeit(lambda: ops.to_nhwc(hm, dt, 32))
    print('k%ds%d: plain %.1f  +stats %.1f | cat %.1f  cat+stats %.1f | pw_k2c_stats(res) %.1f  to_nhwc32 %.1f  [CAT_TILE=%s]' %
          (k, s, t_plain, t_stats, t_cat, t_cats, t_k2c, t_cvt, os.environ.get('MI355_CAT_TILE', '0')))
