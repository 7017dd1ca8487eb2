"""Which Python call sites issue the device-to-device copies and the stock ATen element-wise kernels of one eager A+B+C iteration
(ResNet-50, 256x256, B=64, bf16)?  torch.profiler with stacks; groups the launches by (kernel family, innermost repo frame).
usage: python profiles/find_small_launches.py [batch]"""
import collections
import sys
import torch
sys.path[:0] = ['/root/repo', '/root/repo/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355.da_step import build_training
import uda.model as models
from uda.model.pose_resnet2 import Upsampling
from uda.model.regda_7 import PoseResNetx9
from utils.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0'); mi355.load(); mi355.set_compute_dtype('bf16'); torch.manual_seed(1)
bb = models.resnet50(pretrained=False)
model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
step, opts, scheds = build_training(model, heatmap_size=64)
batch = make_batch(B, 256, 64, seed=1, device=dev)
for _ in range(3):
    step.run(batch)
    for s in scheds.values():
        s.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step.run(batch)
    for s in scheds.values():
        s.step()
    torch.cuda.synchronize()

WATCH = ('aten::copy_', 'aten::clone', 'aten::mul', 'aten::mul_', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_',
         'aten::zeros', 'aten::zeros_like', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::div', 'aten::div_',
         'aten::sum', 'aten::mean', 'aten::ones_like', 'aten::full', 'aten::sub', 'aten::neg', 'aten::detach_')
count = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    # only top-level ops (an aten::clone contains an aten::copy_: count the outermost one with a repo frame)
    if ev.cpu_parent is not None and ev.cpu_parent.name in WATCH:
        continue
    if ev.device_time_total <= 0:          # host-only op (views, metadata)
        continue
    stack = ev.stack or []
    frame = next((f for f in stack if '/repo/' in f or 'hand-pose' in f), stack[0] if stack else '?')
    count[(ev.name, frame.split('/repo/')[-1] + '  ' + str(ev.input_shapes)[:60])] += 1
for (name, frame), n in count.most_common(60):
    print('%4d  %-18s %s' % (n, name, frame))
print('total watched launches per iteration:', sum(count.values()))
mc = collections.Counter()
for ev in prof.events():
    if 'emcpy' in ev.name or 'emset' in ev.name:
        par = ev.cpu_parent
        chain = []
        while par is not None and len(chain) < 4:
            chain.append(par.name); par = par.cpu_parent
        stack = ev.stack or []
        frame = next((f for f in stack if '/repo/' in f or 'hand-pose' in f), '?')
        mc[(ev.name[:40], ' < '.join(chain), frame.split('/repo/')[-1])] += 1
for k, n in mc.most_common(40):
    print('%4d  %s' % (n, k))
