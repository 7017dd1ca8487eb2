"""Is a conv slower right behind its producer than in a loop of itself?  T(graph of n x [producer, conv]) against
T(n x producer) + T(n x conv); producer variants: the BatchNorm apply that writes the conv's input / the same kernel on an unrelated
tensor (code and cache pollution only) / a tiny unrelated kernel."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd', R + '/profiles']
import mi355
from mi355 import ops
mi355.load(); dev = torch.device('cuda:0'); dt = torch.bfloat16
torch.manual_seed(0)

def gtime(fn, n=20, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3

for (N, H, Ci, Co, k) in [(64, 16, 256, 256, 3), (64, 32, 256, 256, 3), (64, 32, 128, 128, 3), (64, 16, 1024, 256, 1), (64, 64, 64, 256, 1), (64, 64, 256, 256, 3)]:
    desc = ops.make_desc(N, H, H, Ci, Co, k, k, 1, k // 2, dt)
    wm = (torch.randn(Co, k, k, Ci) * 0.05).to(dev)
    wf, wt = ops.pack_weights(wm, Co, k * k, Ci, Ci, dt)
    src = ops.nhwc_empty(N, Ci, H, H, dt, dev).normal_()
    other = ops.nhwc_empty(N, Ci, H, H, dt, dev).normal_()
    g = torch.ones(Ci, device=dev); b = torch.zeros(Ci, device=dev); rm = torch.zeros(Ci, device=dev); rv = torch.ones(Ci, device=dev)
    x_fixed = ops.bn_eval_fwd(src, None, g, b, rm, rv, 1e-5, 1)
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    t_conv = gtime(lambda: ops.conv_fwd(desc, x_fixed, wf))
    t_prod = gtime(lambda: ops.bn_eval_fwd(src, None, g, b, rm, rv, 1e-5, 1))
    t_chain = gtime(lambda: ops.conv_fwd(desc, ops.bn_eval_fwd(src, None, g, b, rm, rv, 1e-5, 1), wf))
    def unrelated():
        ops.bn_eval_fwd(other, None, g, b, rm, rv, 1e-5, 1); return ops.conv_fwd(desc, x_fixed, wf)
    t_unrel = gtime(unrelated)
    t_fill = gtime(lambda: big.zero_())
    def flushed():
        big.zero_(); return ops.conv_fwd(desc, x_fixed, wf)
    t_flush = gtime(flushed)
    print('%dx%d %d->%d @%d: conv alone %.1f us | behind its producer %.1f | behind the same kernel on another tensor %.1f | behind a 512 MB fill %.1f   (producer %.1f us)'
          % (k, k, Ci, Co, H, t_conv, t_chain - t_prod, t_unrel - t_prod, t_flush - t_fill, t_prod), flush=True)
