"""Per-parameter weight-gradient agreement of 'fp8' mode with fp8 weight gradients on / off, and of both with bf16 mode."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import nn as mnn
import uda.model as models
from uda.model.pose_resnet2 import Upsampling
from uda.model.regda_7 import PoseResNetx9
dev = torch.device('cuda:0'); mi355.load()
torch.manual_seed(1)
bb = models.resnet50(pretrained=False)
model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
model.train()
x = torch.randn(16, 3, 256, 256, device=dev)
tgt = torch.rand(16, 21, 64, 64, device=dev)
def run(dtype, wg8):
    mi355.set_compute_dtype(dtype); mnn._FP8_WGRAD = wg8
    for p in model.parameters(): p.grad = None
    torch.manual_seed(5)
    out = model(x)
    outs = out if isinstance(out, (tuple, list)) else [out]
    loss = sum(((o.float() - tgt) ** 2).mean() for o in outs if torch.is_tensor(o) and o.shape == tgt.shape)
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None and p.dim() == 4}, float(loss)
run('fp8', False)                     # warm-up: creates the scaling states (first pass is just-in-time scaled)
g16, l16 = run('bf16', True)
g8a, l8a = run('fp8', False)
g8b, l8b = run('fp8', True)
print('loss bf16 %.5f fp8 %.5f fp8+wgrad8 %.5f' % (l16, l8a, l8b))
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
rows = []
for n in g16:
    rows.append((rel(g8b[n], g16[n]), rel(g8a[n], g16[n]), rel(g8b[n], g8a[n]), n, tuple(g16[n].shape)))
rows.sort(key=lambda r: -r[2])
print('%-60s %-22s %10s %10s %10s' % ('parameter', 'shape', 'wg8 vs bf16', 'fp8 vs bf16', 'wg8 vs fp8'))
for a, b, c, n, sh in rows[:25]:
    print('%-60s %-22s %10.3f %10.3f %10.3f' % (n, sh, a, b, c))
import statistics
print('median wg8-vs-bf16 %.3f  fp8-vs-bf16 %.3f' % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))
