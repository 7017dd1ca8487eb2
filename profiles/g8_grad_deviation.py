"""Distribution of the G8 step-A gradient-norm deviations (ours vs the reference's fp64 run), against the per-parameter bound."""
import os, sys
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'domain-adaptative-hand-pose-estimation_amd')
from conftest import golden
import test_gpu_model as T
from mi355.da_step import build_training
import mi355; mi355.set_compute_dtype('f32')
gpu = torch.device('cuda:0')
g = golden('g8_bottleneck')
model = T._g8_setup(gpu); batch = T._g8_batch(gpu)
model.gl_layer.iter_num = 500
step, opts, scheds = build_training(model)
step.skip = True
step._fwdbwd_A(batch); torch.cuda.synchronize()
grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
n32, n64 = g['gradA_norm'], g['gradA_norm64']
rows = []
for k, a32, a64 in zip(g['gradA_keys'], n32, n64):
    if a64 < 1e-6 * n64.max():
        continue
    n = float(grads[str(k)].norm())
    rows.append((abs(n - a64) / (3 * abs(a32 - a64) + 3e-3 * a64), abs(n - a64) / a64, abs(a32 - a64) / a64, str(k)))
rows.sort(reverse=True)
mine = np.array([r[1] for r in rows]); ref = np.array([r[2] for r in rows])
print('S2D', os.environ.get('MI355_STEM_S2D', '1'), 'params', len(rows), 'median mine %.2e ref %.2e | max mine %.2e ref %.2e | over bound: %d' % (
    np.median(mine), np.median(ref), mine.max(), ref.max(), sum(r[0] > 1 for r in rows)))
for r in rows[:6]:
    print('  ratio-to-bound %.2f  mine %.2e  ref32 %.2e  %s' % r)

# the three losses of one complete A/B/C iteration (B and C run on the weights step A updated) against the reference's fp64 run

model2 = T._g8_setup(gpu)
model2.train()
model2.gl_layer.iter_num = 500
step, opts, scheds = build_training(model2)
step.skip = True
out = step.run(batch)
got = [float(out['loss_s']), float(out['loss_gf']), float(out['loss_gt'])]
for i, name in enumerate(('loss_s', 'loss_gf', 'loss_gt')):
    l32, l64 = float(g['losses'][i]), float(g['losses64'][i])
    print('  %s  mine %.6f  ref32 %.6f  ref64 %.6f | rel distance to fp64: mine %.2e  ref32 %.2e' % (
        name, got[i], l32, l64, abs(got[i] - l64) / abs(l64), abs(l32 - l64) / abs(l64)))
