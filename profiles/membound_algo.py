"""Algorithmic bytes per training iteration of the memory-bound kernel families (SURVEY 8(d) per-unit figures x the units
one iteration processes), recorded by hooking the op wrappers during one eager iteration of the benchmark configuration:
BN forward 3 (2 when the conv fused the statistics) passes, BN backward 5 passes, KL = pred + target read + gradient
written, arg-max / soft-arg-max = every element once, pseudo-label = coordinates read + maps written, conv family = the
library's own count.   usage: python profiles/membound_algo.py <iterations_in_trace> <command string> > algo.json"""
import collections
import json
import subprocess
import sys

import torch

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
B, S = 64, 256
bb = models.resnet50(pretrained=False)
model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
step, opts, scheds = build_training(model, heatmap_size=S // 4)
for c in step.crit.values():
    if hasattr(c, 'guard_empty_maps'):
        c.guard_empty_maps = True
batch = make_batch(B, S, S // 4, seed=1, device=dev)
for _ in range(2):
    step.run(batch)
acc = collections.Counter()
nb = lambda t: t.numel() * t.element_size()
o = {n: getattr(ops, n) for n in ('bn_train_fwd', 'bn_bwd', 'kl_heatmap', 'argmax2d', 'softargmax', 'pseudo_label')}


def bn_fwd(x, residual, *a, **k):
    acc['bn_fwd'] += nb(x) * ((2 if k.get('partial') is not None else 3) + (1 if residual is not None else 0))
    return o['bn_train_fwd'](x, residual, *a, **k)


def bn_bwd(dy, x, y, *a, **k):
    want_dres = a[7] if len(a) > 7 else k.get('want_dres', False)
    acc['bn_bwd'] += nb(x) * (5 + (1 if y is not None else 0) + (1 if want_dres else 0))
    return o['bn_bwd'](dy, x, y, *a, **k)


def kl(pred, target, weight, eps, want_grad, *a, **k):
    acc['kl_loss'] += pred.numel() * 4 * (3 if want_grad else 2)
    return o['kl_heatmap'](pred, target, weight, eps, want_grad, *a, **k)


def am(hm):
    acc['argmax'] += hm.numel() * 4
    return o['argmax2d'](hm)


def pl(xy, patch, radius, div, S_, kind, extra=None, normalise=False, want_gt=True, want_gf=True):
    n = xy.shape[0] * xy.shape[1] * S_ * S_ * 4
    acc['pseudo_label'] += n * (int(want_gt) + int(want_gf) + (1 if extra is not None else 0))
    return o['pseudo_label'](xy, patch, radius, div, S_, kind, extra, normalise, want_gt, want_gf)


ops.bn_train_fwd, ops.bn_bwd, ops.kl_heatmap, ops.argmax2d, ops.pseudo_label = bn_fwd, bn_bwd, kl, am, pl
ops.prof_reset(); ops.prof_enable(True)
step.run(batch); torch.cuda.synchronize()
ops.prof_enable(False)
ms, launches, flops, abytes = ops.prof_read()
fam = {k: {'bytes_per_iteration': int(v)} for k, v in acc.items()}
fam['conv_mfma'] = {'bytes_per_iteration': int(abytes), 'launches_per_iteration': int(launches), 'flops_per_iteration': flops}
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd='/root/repo').stdout.strip()
if not head:      # the GPU box receives a snapshot without .git: the commit is written next to this script before the run
    try:
        head = open('/root/repo/profiles/.head_for_pmc').read().strip()
    except OSError:
        head = ''
print(json.dumps({'iterations_in_trace': int(sys.argv[1]), 'command': sys.argv[2] if len(sys.argv) > 2 else None, 'git_head': head or None,
                  'config': 'resnet50_256_b64_bf16', 'families': fam}, indent=1))
