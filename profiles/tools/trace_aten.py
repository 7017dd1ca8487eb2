"""Which Python lines launch ATen kernels inside one eager A+B+C iteration (TorchDispatchMode + stack)."""
import sys, os, collections, traceback, torch
R = os.getcwd(); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355.da_step import build_training
import uda.model as models
from uda.model.pose_resnet2 import Upsampling
from uda.model.regda_7 import PoseResNetx9
from utils.synthetic import make_batch
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0'); mi355.load(); mi355.set_compute_dtype('bf16')
torch.manual_seed(1)
bb = models.resnet50(pretrained=False)
model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
step, opts, scheds = build_training(model, heatmap_size=64)
batch = make_batch(8, 256, 64, seed=1, device=dev)
for _ in range(3): step.run(batch)
torch.cuda.synchronize()
cnt = collections.Counter()
SKIP = ('aten.empty', 'aten.view', 'aten.as_strided', 'aten.detach', 'aten.alias', 'aten.permute', 'aten.reshape', 'aten._unsafe_view',
        'aten.select', 'aten.slice', 'aten.expand', 'aten.t.', 'aten.transpose', 'aten.unsqueeze', 'aten.squeeze', 'aten.new_empty', 'aten.empty_strided',
        'aten.is_', 'aten.sym_', 'aten._local_scalar', 'aten.stride', 'aten.size', 'aten.lift_fresh', 'aten.unbind', 'aten.split', 'aten.narrow')
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = [f for f in traceback.extract_stack() if 'domain-adaptative' in f.filename and 'trace_aten' not in f.filename]
            where = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in st[-3:][::-1]) if st else '?'
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with M():
    step.run(batch)
torch.cuda.synchronize()
tot = 0
for (n, w), c in cnt.most_common(80):
    print('%3d %-28s %s' % (c, n, w)); tot += c
print('total', tot)
