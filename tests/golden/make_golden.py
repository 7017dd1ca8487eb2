"""Generate golden vectors by importing the reference's live classes (run in the build
container only: needs /root/reference).  Writes tests/golden/*.npz.

Import recipe: the reference's package __init__ files pull in
torchvision / cv2 which are not installed, so `utils`, `uda`, `uda.model` are
pre-registered as namespace stubs pointing at the reference directories and
`uda.model.resnet` (pure torchvision glue) is a dummy; numpy aliases removed in
numpy>=1.24 are restored.  No reference source is copied: only arrays are stored.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root (for oracle backbone in G7)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
np.int = int
np.float = float


def _stub(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub('utils', f'{REF}/utils')
_stub('uda', f'{REF}/uda')
_stub('uda.model', f'{REF}/uda/model')
_stub('uda.model.resnet', _resnet=None, Bottleneck=None)

import utils.gl as ref_gl  # noqa: E402
import utils.keypoint_detection as ref_kd  # noqa: E402
import uda.model.loss as ref_loss  # noqa: E402
import uda.model.regda_4 as ref_r4  # noqa: E402
import uda.model.regda_7 as ref_r7  # noqa: E402
import uda.model.pose_resnet2 as ref_pr2  # noqa: E402
from seeded import fill_module_, randn, rand, peaky_heatmaps, weights_bk, g9_inputs  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, {k: v.shape for k, v in out.items()})


class _Feat(nn.Module):  # stands in for the backbone: identity with .out_features
    def __init__(self, c):
        super().__init__()
        self.out_features = c

    def forward(self, x):
        return x


def g1_neck_heads():
    C = 64
    m = ref_r7.PoseResNetx9(_Feat(C), ref_pr2.Upsampling(C), 256, 21)
    fill_module_(m, 101)
    x = randn(102, 2, C, 8, 8)
    m.train()
    y, y_adv, y_adv2, y_adv3, f = m(x)
    bn = {k.replace('.', '_'): v.clone() for k, v in m.state_dict().items() if 'running' in k and
          (k.startswith('upsampling.1.') or k.startswith('head_adv3.last_lay.6.'))}
    m.eval()
    y_eval = m(x)
    save('g1_neck_heads', y=y, y_adv=y_adv, y_adv2=y_adv2, y_adv3=y_adv3, f_sum=f.sum(), f_abs=f.abs().sum(),
         f_slice=f[:, :4, :8, :8], y_eval=y_eval, **bn)


def _grad(fn, inp):
    inp = inp.clone().requires_grad_(True)
    v = fn(inp)
    v.backward()
    return v.detach(), inp.grad.detach()


def g2_losses():
    B, K = 2, 21
    y = peaky_heatmaps(201, B, K, 64, 64)
    y_adv = randn(202, B, K, 64, 64)
    y_adv2 = randn(203, B, K, 32, 32)
    y_adv3 = randn(204, B, K, 16, 16)
    label = rand(205, B, K, 64, 64) * (rand(206, B, K, 64, 64) > 0.9)
    w = weights_bk(207, B, K)
    up = lambda t, s: nn.Upsample(size=s, mode='bilinear')(t)
    target5 = 0.5 * up(y_adv3, 64) + up(y_adv2, 64)
    target0 = up(y_adv3, 32)
    kl0, kl7 = ref_loss.JointsKLLoss(), ref_loss.JointsKLLoss(epsilon=1e-7)
    rd6 = ref_r7.RegressionDisparityx6(ref_r4.PseudoLabelGenerator(K, 64, 64), ref_loss.JointsKLLoss(epsilon=1e-7))
    rd5 = ref_r7.RegressionDisparityx5(ref_r7.PseudoLabelGenerator03(K), ref_loss.JointsKLLoss(epsilon=1e-7))
    rd1 = ref_r7.RegressionDisparityx1(ref_r7.PseudoLabelGenerator01(K), ref_loss.JointsKLLoss(epsilon=1e-7))
    out = dict(target5=target5[:, :2], target0=target0[:, :2])
    for name, fn, inp in [
        ('kl0', lambda p: kl0(p, label, w), y_adv),
        ('kl7', lambda p: kl7(p, label, w), y_adv),
        ('kl0_now', lambda p: kl0(p, label), y_adv),
        ('x1_min', lambda p: rd1(y, p, w, mode='min'), y_adv3),
        ('x1_max', lambda p: rd1(y, p, w, mode='max'), y_adv3),
        ('x5_min', lambda p: rd5(y, p, None, w, mode='min'), y_adv2),
        ('x5_max_none', lambda p: rd5(y, p, None, w, mode='max'), y_adv2),
        ('x5_max_t0', lambda p: rd5(y, p, target0, w, mode='max'), y_adv2),
        ('x6_min', lambda p: rd6(y, p, None, w, mode='min'), y_adv),
        ('x6_max_none', lambda p: rd6(y, p, None, w, mode='max'), y_adv),
        ('x6_max_t5', lambda p: rd6(y, p, target5, w, mode='max'), y_adv),
    ]:
        v, g = _grad(fn, inp)
        out[name] = v
        out[name + '_grad'] = g[:, ::5]  # every 5th keypoint keeps the fixture small
        out[name + '_gsum'] = g.double().abs().sum()
    save('g2_losses', **out)


def g3_pseudo_labels():
    B, K = 2, 21
    y = peaky_heatmaps(301, B, K, 64, 64)
    gt, gf = ref_r4.PseudoLabelGenerator(K, 64, 64)(y)
    gt1, gf1 = ref_r7.PseudoLabelGenerator01(K)(y)
    gt3, gf3 = ref_r7.PseudoLabelGenerator03(K)(y)
    save('g3_pseudo_labels', gt=gt, gf=gf, gt01=gt1, gf01=gf1, gt03=gt3, gf03=gf3)


def g4_argmax_accuracy():
    B, K = 3, 21
    hm = peaky_heatmaps(401, B, K, 64, 64).numpy()
    hm[1, 0] = 0.5  # whole-map tie -> first index
    hm[1, 1, 10, 7] = hm[1, 1, 40, 3] = 9.0  # two-way tie -> lower flat index
    hm[2, 2, 63, 63] = 11.0
    lab = peaky_heatmaps(402, B, K, 64, 64).numpy()
    lab[:, :, :, :] = np.maximum(lab, 0)
    preds, maxvals = ref_kd.get_max_preds(hm)
    acc, avg, cnt, pred = ref_kd.accuracy(hm, lab)
    save('g4_argmax_accuracy', preds=preds, maxvals=maxvals, acc=acc, avg=avg, cnt=cnt, pred=pred)


def g5_softargmax():
    hm = randn(501, 2, 21, 64, 64, scale=0.05)
    hm[0, 0, 20, 33] += 1.0
    save('g5_softargmax', uv=ref_kd.compute_uv_from_heatmaps3(hm))


def g6_gl():
    its = [0, 1, 100, 1000, 10000, 100000]
    lam = []
    for i in its:
        layer = ref_gl.WarmStartGradientLayer(alpha=1.0, lo=0.0, hi=0.1, max_iters=1000, auto_step=False)
        layer.iter_num = i
        x = torch.ones(4, requires_grad=True)
        layer(x).sum().backward()
        lam.append(float(x.grad[0]))
    save('g6_gl', iters=np.array(its), lam=np.array(lam, dtype=np.float64))


def g7_iteration():
    """Steps A/B/C of train1.py:371-458 driven over the reference's own model / loss classes
    (backbone: oracle torchvision-layout ResNet-18, since torchvision is absent)."""
    from torch.optim import SGD
    from torch.optim.lr_scheduler import LambdaLR
    from oracle.backbone import make_backbone
    bb = make_backbone('resnet18')
    model = ref_r7.PoseResNetx9(bb, ref_pr2.Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    fill_module_(model, 701)
    B = 2
    x_s, x_t = randn(702, B, 3, 256, 256), randn(703, B, 3, 256, 256)
    label_s = rand(704, B, 21, 64, 64) * (rand(705, B, 21, 64, 64) > 0.9)
    w_s, w_t = weights_bk(706, B, 21), weights_bk(707, B, 21)
    criterion = ref_loss.JointsKLLoss()
    kl = lambda: ref_loss.JointsKLLoss(epsilon=1e-7)
    rd = ref_r7.RegressionDisparityx6(ref_r4.PseudoLabelGenerator(21, 64, 64), kl())
    rd2 = ref_r7.RegressionDisparityx5(ref_r7.PseudoLabelGenerator03(21), kl())
    rd1 = ref_r7.RegressionDisparityx1(ref_r7.PseudoLabelGenerator01(21), kl())
    mk = lambda ps: SGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    of = mk([{'params': bb.parameters(), 'lr': 0.1}, {'params': model.upsampling.parameters(), 'lr': 0.1}])
    oh, oa, oa2, oa3 = (mk(getattr(model, n).parameters()) for n in ('head', 'head_adv', 'head_adv2', 'head_adv3'))
    opts = [of, oh, oa, oa2, oa3]
    scheds = [LambdaLR(o, lambda x: 0.01 * (1. + 1e-4 * float(x)) ** (-0.75)) for o in opts]
    model.gl_layer.iter_num = 500  # mid-schedule lambda so the GL path matters
    model.train()
    res = {}
    for it in range(2):
        for o in opts:
            o.zero_grad()
        y_s, y_s_adv, y_s_adv2, y_s_adv3, f_s = model(x_s)
        loss_s = 2 * criterion(y_s, label_s, w_s) + 4 * rd2(y_s, y_s_adv2, None, w_s, mode='min') + \
            4 * rd(y_s, y_s_adv, None, w_s, mode='min') + 4 * rd1(y_s, y_s_adv3, w_s, mode='min')
        loss_s.backward()
        for o in opts:
            o.step()
        for o in (oa, oa2, oa3):
            o.zero_grad()
        y_t, y_t_adv, y_t_adv2, y_t_adv3, f_t = model(x_t)
        l1 = rd1(y_t, y_t_adv3, w_t, mode='max')
        t = nn.Upsample(size=64, mode='bilinear')(y_t_adv3.detach())
        t1 = nn.Upsample(size=64, mode='bilinear')(y_t_adv2.detach())
        t0 = nn.Upsample(size=32, mode='bilinear')(y_t_adv3.detach())
        l2 = rd(y_t, y_t_adv, 0.5 * t + t1, w_t, mode='max')
        l3 = rd2(y_t, y_t_adv2, t0, w_t, mode='max')
        loss_gf = 0.3 * l1 + 1 * l2 + 0.3 * l3
        loss_gf.backward()
        oa2.step(); oa.step(); oa3.step()
        of.zero_grad()
        y_t, y_t_adv, y_t_adv2, y_t_adv3, f_t = model(x_t)
        loss_gt = 0.3 * rd2(y_t, y_t_adv2, None, w_t, mode='min') + 1 * rd(y_t, y_t_adv, None, w_t, mode='min')
        loss_gt.backward()
        of.step()
        model.step()
        for s in scheds:
            s.step()
        res[f'it{it}_losses'] = np.array([float(loss_s), float(loss_gf), float(loss_gt)], dtype=np.float64)
        if it == 0:
            res['it0_y_s'] = y_s[:, ::5].detach().clone()
    sd = model.state_dict()
    keys = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    res['param_sum'] = np.array([float(sd[k].double().sum()) for k in keys])
    res['param_abs'] = np.array([float(sd[k].double().abs().sum()) for k in keys])
    res['param_keys'] = np.array(keys)
    save('g7_iteration', **res)


def _ref_da_model(arch, seed):
    from oracle.backbone import make_backbone
    bb = make_backbone(arch)
    model = ref_r7.PoseResNetx9(bb, ref_pr2.Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    fill_module_(model, seed)
    return bb, model


def _g8_iteration(dt):
    from torch.optim import SGD
    from torch.optim.lr_scheduler import LambdaLR
    bb, model = _ref_da_model('resnet50', 801)
    model = model.to(dt)
    B = 2
    x_s, x_t = randn(802, B, 3, 256, 256).to(dt), randn(8034, B, 3, 256, 256).to(dt)
    label_s = (rand(804, B, 21, 64, 64) * (rand(805, B, 21, 64, 64) > 0.9)).to(dt)
    w_s, w_t = weights_bk(806, B, 21).to(dt), weights_bk(807, B, 21).to(dt)
    criterion = ref_loss.JointsKLLoss()
    kl = lambda: ref_loss.JointsKLLoss(epsilon=1e-7)
    rd = ref_r7.RegressionDisparityx6(ref_r4.PseudoLabelGenerator(21, 64, 64), kl())
    rd2 = ref_r7.RegressionDisparityx5(ref_r7.PseudoLabelGenerator03(21), kl())
    rd1 = ref_r7.RegressionDisparityx1(ref_r7.PseudoLabelGenerator01(21), kl())
    mk = lambda ps: SGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    of = mk([{'params': bb.parameters(), 'lr': 0.1}, {'params': model.upsampling.parameters(), 'lr': 0.1}])
    oh, oa, oa2, oa3 = (mk(getattr(model, n).parameters()) for n in ('head', 'head_adv', 'head_adv2', 'head_adv3'))
    opts = [of, oh, oa, oa2, oa3]
    scheds = [LambdaLR(o, lambda x: 0.01 * (1. + 1e-4 * float(x)) ** (-0.75)) for o in opts]
    model.gl_layer.iter_num = 500
    model.train()
    res = {}
    for o in opts:
        o.zero_grad()
    y_s, y_s_adv, y_s_adv2, y_s_adv3, f_s = model(x_s)
    res.update(y_s=y_s[:, ::5].detach().clone(), y_s_adv=y_s_adv[:, ::5].detach().clone(),
               y_s_adv2=y_s_adv2[:, ::5].detach().clone(), y_s_adv3=y_s_adv3[:, ::5].detach().clone(),
               f_slice=f_s[:, :8, :16, :16].detach().clone(), f_abs=f_s.detach().double().abs().sum())
    loss_s = 2 * criterion(y_s, label_s, w_s) + 4 * rd2(y_s, y_s_adv2, None, w_s, mode='min') + \
        4 * rd(y_s, y_s_adv, None, w_s, mode='min') + 4 * rd1(y_s, y_s_adv3, w_s, mode='min')
    loss_s.backward()
    gkeys = sorted(k for k, p in model.named_parameters() if p.grad is not None)
    res['gradA_keys'] = np.array(gkeys)
    gp = dict(model.named_parameters())
    res['gradA_norm'] = np.array([float(gp[k].grad.double().norm()) for k in gkeys])
    res['gradA_sum'] = np.array([float(gp[k].grad.double().sum()) for k in gkeys])
    for o in opts:
        o.step()
    for o in (oa, oa2, oa3):
        o.zero_grad()
    y_t, y_t_adv, y_t_adv2, y_t_adv3, f_t = model(x_t)
    l1 = rd1(y_t, y_t_adv3, w_t, mode='max')
    t = nn.Upsample(size=64, mode='bilinear')(y_t_adv3.detach())
    t1 = nn.Upsample(size=64, mode='bilinear')(y_t_adv2.detach())
    t0 = nn.Upsample(size=32, mode='bilinear')(y_t_adv3.detach())
    l2 = rd(y_t, y_t_adv, 0.5 * t + t1, w_t, mode='max')
    l3 = rd2(y_t, y_t_adv2, t0, w_t, mode='max')
    loss_gf = 0.3 * l1 + 1 * l2 + 0.3 * l3
    loss_gf.backward()
    oa2.step(); oa.step(); oa3.step()
    of.zero_grad()
    y_t, y_t_adv, y_t_adv2, y_t_adv3, f_t = model(x_t)
    loss_gt = 0.3 * rd2(y_t, y_t_adv2, None, w_t, mode='min') + 1 * rd(y_t, y_t_adv, None, w_t, mode='min')
    loss_gt.backward()
    of.step()
    model.step()
    for s in scheds:
        s.step()
    res["losses"] = np.array([float(v.detach()) for v in (loss_s, loss_gf, loss_gt)], dtype=np.float64)
    sd = model.state_dict()
    keys = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    res['param_sum'] = np.array([float(sd[k].double().sum()) for k in keys])
    res['param_abs'] = np.array([float(sd[k].double().abs().sum()) for k in keys])
    res['param_keys'] = np.array(keys)
    res['nbt_layer4'] = np.array(int(sd['backbone.layer4.2.bn3.num_batches_tracked']))
    return res


def g8_bottleneck():
    """The benchmarked architectures (Bottleneck nets, uda/model/resnet.py:92-107): one complete A/B/C iteration of
    train1.py:371-458 over the reference's PoseResNetx9 + loss classes on a ResNet-50 layout (B=2, 256x256, fp32):
    5-tuple slices of the step-A forward, per-parameter gradient norms of step A, the three losses, every parameter's
    sum and |sum| after the iteration.  The same run in fp64 (keys *64) is the yardstick for the quantities that sit
    behind an SGD update: gradients of this random-init network are ill-conditioned (fp32 vs fp64 of the reference's
    own classes: ~1e-2 relative per parameter), so losses of steps B / C differ by ~1e-3 between any two
    implementations.  Plus a ResNet-101 train-mode forward (slices + BN running statistics of the deepest stages)."""
    res = _g8_iteration(torch.float32)
    r64 = _g8_iteration(torch.float64)
    assert list(r64['gradA_keys']) == list(res['gradA_keys'])
    res.update(losses64=r64['losses'], gradA_norm64=r64['gradA_norm'], param_abs64=r64['param_abs'])
    B = 2
    # ResNet-101: train-mode forward only; the fp64 run of the same classes is the yardstick (B=2 batch statistics over
    # 128 samples per channel in layer4 make the deep train-mode forward itself sensitive: fp32 vs fp64 2e-3 at y_adv3)
    for dt, sfx in ((torch.float32, ''), (torch.float64, '64')):
        bb, m101 = _ref_da_model('resnet101', 811)
        m101 = m101.to(dt).train()
        x = randn(812, B, 3, 256, 256).to(dt)
        with torch.no_grad():
            y, y_adv, y_adv2, y_adv3, f = m101(x)
        sd = m101.state_dict()
        res.update({'r101_y' + sfx: y[:, ::5].clone(), 'r101_y_adv3' + sfx: y_adv3[:, ::5].clone(),
                    'r101_f_slice' + sfx: f[:, :8, :16, :16].clone(), 'r101_f_abs' + sfx: f.double().abs().sum(),
                    'r101_rm' + sfx: sd['backbone.layer3.22.bn3.running_mean'].clone(),
                    'r101_rv' + sfx: sd['backbone.layer4.2.bn3.running_var'].clone()})
        if dt == torch.float32:
            m101.eval()
            with torch.no_grad():
                res['r101_y_eval'] = m101(x)[:, ::5].clone()
    save('g8_bottleneck', **res)



def g9_generate_target():
    """uda/dataset/util.py:9-68 run on the reference itself (cv2 / scipy.io are import-only there: cv2 is stubbed)."""
    _stub('cv2')
    _stub('uda.dataset', f'{REF}/uda/dataset')
    import uda.dataset.util as ref_util
    kp, vis = g9_inputs()
    out = {}
    for tag, (hm, img) in dict(a=((64, 64), (256, 256)), b=((32, 32), (128, 128)), c=((128, 128), (512, 512))).items():
        scale = img[0] / 256.0
        t, w = zip(*[ref_util.generate_target(kp[b] * scale, vis[b], hm, 2, img) for b in range(kp.shape[0])])
        out['target_' + tag], out['weight_' + tag] = np.stack(t), np.stack(w)
    save('g9_generate_target', **out)



def g10_dataset_util():
    """Geometry helpers of the data layer run on the reference (uda/dataset/util.py:72-143; cv2 stubbed, import only)."""
    _stub('cv2')
    _stub('uda.dataset', f'{REF}/uda/dataset')
    import uda.dataset.util as ref_util
    rng = np.random.default_rng(1001)
    boxes = rng.uniform(-40, 360, size=(200, 4))
    boxes[:, 2] = boxes[:, 0] + np.abs(rng.normal(60, 50, 200)); boxes[:, 3] = boxes[:, 1] + np.abs(rng.normal(60, 50, 200))
    dims = rng.integers(100, 700, size=(200, 2))
    scales = rng.choice([1.0, 1.5, 1.6, 2.0], size=200)
    scaled = np.array([ref_util.scale_box(tuple(b), int(w), int(h), float(s)) for b, (w, h), s in zip(boxes, dims, scales)], dtype=np.float64)
    kp = rng.uniform(0, 320, size=(10, 21, 2))
    Kmat = np.array([[283.1, 0, 160.0], [0, 283.1, 160.0], [0, 0, 1.0]])
    Zc = rng.uniform(0.3, 1.2, size=(10, 21))
    xyz = np.stack([ref_util.keypoint2d_to_3d(kp[i], Kmat, Zc[i]) for i in range(10)])
    uv = np.stack([ref_util.keypoint3d_to_2d(xyz[i], Kmat) for i in range(10)])
    bbox = np.array([ref_util.get_bounding_box(kp[i]) for i in range(10)])
    ia = boxes[:100].round(); ib = boxes[100:].round()
    inter = np.array([ref_util.intersection(tuple(a), tuple(b)) for a, b in zip(ia, ib)])
    areas = np.array([ref_util.area(*t) for t in inter])
    save('g10_dataset_util', boxes=boxes, dims=dims, scales=scales, scaled=scaled, kp=kp, K=Kmat, Zc=Zc, xyz=xyz, uv=uv, bbox=bbox,
         inter=inter, areas=areas)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10']
    fns = {'g1': g1_neck_heads, 'g2': g2_losses, 'g3': g3_pseudo_labels, 'g4': g4_argmax_accuracy,
           'g5': g5_softargmax, 'g6': g6_gl, 'g7': g7_iteration, 'g8': g8_bottleneck, 'g9': g9_generate_target, 'g10': g10_dataset_util}
    for w in which:
        fns[w]()
