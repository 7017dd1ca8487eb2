import sys, os, torch, collections
R = os.getcwd(); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops
import uda.model as models
from uda.model.pose_resnet2 import Upsampling
from uda.model.regda_7 import PoseResNetx9
from utils.synthetic import make_batch
dev = torch.device('cuda:0'); mi355.load(); mi355.set_compute_dtype('bf16')
torch.manual_seed(1)
bb = models.resnet50(pretrained=False)
model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
batch = make_batch(64, 256, 64, seed=1, device=dev)
model.eval()
ov = ops.prof_event_overhead_us(256)
with torch.no_grad():
    for _ in range(3): model(batch['x_t'])
    torch.cuda.synchronize()
    ops.prof_reset(); ops.prof_enable(1)
    for _ in range(5):
        ops.spin_us(30000); model(batch['x_t']); torch.cuda.synchronize()
    ops.prof_enable(0)
L = ops.prof_launches()
agg = collections.OrderedDict()
for e in L:
    a = agg.setdefault(e['label'], [0, 0.0, e['flops'], e['bytes']])
    a[0] += 1; a[1] += max(e['us'] - ov, 0)
tot = 0
rows = []
for lab, (n, us, fl, by) in agg.items():
    m = us / n; per = n / 5.0
    floor = max(fl / 1.75e15 * 1e6, by / 6.3e12 * 1e6, 5.0)
    rows.append(((m - floor) * per, lab, per, m, floor, fl / m / 1e6)); tot += m * per
rows.sort(reverse=True)
print('eval forward: conv family %.3f ms per batch of 64 (event-timed, overhead %.1f us subtracted)' % (tot / 1e3, ov))
for gap, lab, per, m, floor, tf in rows[:40]:
    print('%-50s n=%4.1f  %7.1f us  floor %6.1f  gap %.3f ms  %5.0f TF/s' % (lab, per, m, floor, gap / 1e3, tf))
