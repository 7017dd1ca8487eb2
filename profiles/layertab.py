"""Per-layer GEMM table: record every conv launch of one eager training iteration, then time each unique
(kind, shape) in isolation.  Shows where the conv family's time goes."""
import sys, os, collections, torch
sys.path[:0] = ['/root/repo', '/root/repo/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops
from mi355.da_step import build_training
import uda.model as models
from uda.model.pose_resnet2 import Upsampling
from uda.model.regda_7 import PoseResNetx9
from utils.synthetic import make_batch
dev = torch.device('cuda:0'); mi355.load(); mi355.set_compute_dtype('bf16')
torch.manual_seed(1)
arch = os.environ.get('ARCH', 'resnet50'); S = int(os.environ.get('IMG', '256')); B = int(os.environ.get('BATCH', '64'))
bb = models.__dict__[arch](pretrained=False)
model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
step, opts, scheds = build_training(model, heatmap_size=S // 4)
batch = make_batch(B, S, S // 4, seed=1, device=dev)
for _ in range(2): step.run(batch)
log = collections.Counter()
F = ('N', 'Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'kh', 'kw', 'stride', 'pad')
key = lambda d: tuple(getattr(d, f) for f in F)
o_f, o_d, o_w = ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad
def f1(desc, *a, **k): log[('fwd', key(desc))] += 1; return o_f(desc, *a, **k)
def f2(desc, *a, **k): log[('dgrad', key(desc))] += 1; return o_d(desc, *a, **k)
def f3(desc, *a, **k): log[('wgrad', key(desc))] += 1; return o_w(desc, *a, **k)
ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = f1, f2, f3
o_fs, o_ds = ops.conv_fwd_stats, ops.conv_dgrad_stats
def f4(desc, *a, **k): log[('fwd', key(desc))] += 1; return o_fs(desc, *a, **k)
def f5(desc, *a, **k): log[('dgrad', key(desc))] += 1; return o_ds(desc, *a, **k)
ops.conv_fwd_stats, ops.conv_dgrad_stats = f4, f5
o_ma = ops.conv_dgrad_masked_acc
def f6(desc, *a, **k): log[('dgrad', key(desc))] += 1; return o_ma(desc, *a, **k)
ops.conv_dgrad_masked_acc = f6
step.run(batch); torch.cuda.synchronize()
ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = o_f, o_d, o_w
ops.conv_fwd_stats, ops.conv_dgrad_stats = o_fs, o_ds
ops.conv_dgrad_masked_acc = o_ma
del step, model, opts; torch.cuda.empty_cache()
dt = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
rows = []
KIND = os.environ.get('KIND')
for (kind, k), cnt in log.items():
    if KIND and kind != KIND: continue
    N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, s, p = k
    desc = ops.make_desc(N, Hi, Wi, Ci, Co, kh, kw, s, p, dt)
    x = ops.nhwc_empty(N, Ci, Hi, Wi, dt, dev).normal_(); dy = ops.nhwc_empty(N, Co, Ho, Wo, dt, dev).normal_()
    w = torch.randn(Co * kh * kw * Ci, device=dev).to(dt); dw = torch.empty(Co * kh * kw * Ci, device=dev)
    fl = 2.0 * N * Ho * Wo * Co * kh * kw * Ci
    if kind == 'fwd': t = timeit(lambda: ops.conv_fwd(desc, x, w))
    elif kind == 'dgrad': t = timeit(lambda: ops.conv_dgrad(desc, dy, w))
    else: t = timeit(lambda: ops.conv_wgrad(desc, x, dy, dw, False))
    rows.append((t * cnt * 1e3, kind, k, cnt, t * 1e6, fl / t / 1e12))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows); fl_tot = sum(r[5] * 1e12 * r[4] * 1e-6 * r[3] for r in rows)
print('%-6s %-46s %4s %9s %7s %8s %6s' % ('kind', 'N Hi Wi Ci Ho Wo Co kh kw s p', 'cnt', 'us', 'TF/s', 'ms/iter', 'cum%'))
cum = 0
for ms, kind, k, cnt, us, tf in rows:
    cum += ms
    print('%-6s %-46s %4d %9.1f %7.0f %8.3f %6.1f' % (kind, ' '.join(map(str, k)), cnt, us, tf, ms, 100 * cum / tot))
print('total %.2f ms/iter, %.0f TFLOP/s average' % (tot, fl_tot / tot / 1e9))
