import sys, os, json, torch, numpy as np
R = os.getcwd(); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd', R + '/tests', R + '/tests/golden']
import mi355
mi355.load(); mi355.set_compute_dtype('f32')
from mi355.da_step import build_training
from test_gpu_model import _g8_setup, _g8_batch
gpu = torch.device('cuda:0')
model = _g8_setup(gpu); batch = _g8_batch(gpu)
model.gl_layer.iter_num = 500
step, opts, scheds = build_training(model)
step.skip = True
out = {}
model.train()
step._begin_reduce(()); 
step._fwdbwd_A(batch); torch.cuda.synchronize()
out['gradA'] = {k: float(p.grad.double().abs().sum()) for k, p in model.named_parameters() if p.grad is not None}
step._update_A(); torch.cuda.synchronize()
out['paramA'] = {k: float(p.double().abs().sum()) for k, p in model.named_parameters()}
out['loss_s'] = float(step.out['loss_s'])
# step B pieces
import mi355 as _rt
with _rt.bn_updates(2):
    f_t = model.features(batch['x_t']); y_t = model.head(f_t).detach()
out['f_t'] = float(f_t.double().abs().sum()); out['y_t'] = float(y_t.double().abs().sum())
y_adv, y_adv2, y_adv3 = model.adv_heads(f_t.detach())
for n, t in (('y_adv', y_adv), ('y_adv2', y_adv2), ('y_adv3', y_adv3)): out[n] = float(t.double().abs().sum())
json.dump(out, open(sys.argv[1], 'w'))
