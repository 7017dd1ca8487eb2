"""Pin the CPU oracle against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  Same ATen ops on the same CPU => tight tolerances."""
import numpy as np
import torch
import torch.nn as nn
from conftest import golden
from seeded import fill_module_, randn, rand, peaky_heatmaps, weights_bk

from oracle import pose as op, losses as ol
from oracle.backbone import make_backbone
from oracle.train_step import DATrainer

torch.set_num_threads(8)
T = dict(rtol=1e-5, atol=1e-6)


class _Feat(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.out_features = c

    def forward(self, x):
        return x


def test_g1_neck_heads():
    g = golden('g1_neck_heads')
    m = op.PoseResNetx9(_Feat(64), op.Upsampling(64), 256, 21)
    fill_module_(m, 101)
    x = randn(102, 2, 64, 8, 8)
    m.train()
    y, y_adv, y_adv2, y_adv3, f = m(x)
    for name, t in dict(y=y, y_adv=y_adv, y_adv2=y_adv2, y_adv3=y_adv3).items():
        np.testing.assert_allclose(t.detach().numpy(), g[name], **T)
    np.testing.assert_allclose(f[:, :4, :8, :8].detach().numpy(), g['f_slice'], **T)
    sd = m.state_dict()
    for k in ('upsampling.1.running_mean', 'upsampling.1.running_var', 'head_adv3.last_lay.6.running_var'):
        np.testing.assert_allclose(sd[k].numpy(), g[k.replace('.', '_')], **T)
    m.eval()
    np.testing.assert_allclose(m(x).detach().numpy(), g['y_eval'], **T)


def _grad(fn, inp):
    inp = inp.clone().requires_grad_(True)
    v = fn(inp)
    v.backward()
    return v.detach(), inp.grad.detach()


def test_g2_losses():
    g = golden('g2_losses')
    B, K = 2, 21
    y = peaky_heatmaps(201, B, K, 64, 64)
    y_adv, y_adv2, y_adv3 = randn(202, B, K, 64, 64), randn(203, B, K, 32, 32), randn(204, B, K, 16, 16)
    label = rand(205, B, K, 64, 64) * (rand(206, B, K, 64, 64) > 0.9)
    w = weights_bk(207, B, K)
    up = lambda t, s: nn.Upsample(size=s, mode='bilinear')(t)
    target5, target0 = 0.5 * up(y_adv3, 64) + up(y_adv2, 64), up(y_adv3, 32)
    kl0, kl7 = ol.JointsKLLoss(), ol.JointsKLLoss(epsilon=1e-7)
    rd6 = ol.RegressionDisparityx6(ol.PseudoLabelGenerator(K, 64, 64), ol.JointsKLLoss(epsilon=1e-7))
    rd5 = ol.RegressionDisparityx5(ol.PseudoLabelGenerator03(K), ol.JointsKLLoss(epsilon=1e-7))
    rd1 = ol.RegressionDisparityx1(ol.PseudoLabelGenerator01(K), ol.JointsKLLoss(epsilon=1e-7))
    cases = [('kl0', lambda p: kl0(p, label, w), y_adv), ('kl7', lambda p: kl7(p, label, w), y_adv),
             ('kl0_now', lambda p: kl0(p, label), y_adv),
             ('x1_min', lambda p: rd1(y, p, w, mode='min'), y_adv3), ('x1_max', lambda p: rd1(y, p, w, mode='max'), y_adv3),
             ('x5_min', lambda p: rd5(y, p, None, w, mode='min'), y_adv2),
             ('x5_max_none', lambda p: rd5(y, p, None, w, mode='max'), y_adv2),
             ('x5_max_t0', lambda p: rd5(y, p, target0, w, mode='max'), y_adv2),
             ('x6_min', lambda p: rd6(y, p, None, w, mode='min'), y_adv),
             ('x6_max_none', lambda p: rd6(y, p, None, w, mode='max'), y_adv),
             ('x6_max_t5', lambda p: rd6(y, p, target5, w, mode='max'), y_adv)]
    for name, fn, inp in cases:
        v, gr = _grad(fn, inp)
        assert np.isfinite(float(v)), name
        np.testing.assert_allclose(float(v), float(g[name]), rtol=1e-6, err_msg=name)
        np.testing.assert_allclose(gr[:, ::5].numpy(), g[name + '_grad'], rtol=1e-5, atol=1e-9, err_msg=name)


def test_g3_pseudo_labels_bit_exact():
    g = golden('g3_pseudo_labels')
    y = peaky_heatmaps(301, 2, 21, 64, 64)
    for cls, a, b in [(lambda: ol.PseudoLabelGenerator(21, 64, 64), 'gt', 'gf'),
                      (lambda: ol.PseudoLabelGenerator01(21), 'gt01', 'gf01'),
                      (lambda: ol.PseudoLabelGenerator03(21), 'gt03', 'gf03')]:
        gt, gf = cls()(y)
        assert np.array_equal(gt.numpy(), g[a]) and np.array_equal(gf.numpy(), g[b])


def _g4_inputs():
    hm = peaky_heatmaps(401, 3, 21, 64, 64).numpy()
    hm[1, 0] = 0.5
    hm[1, 1, 10, 7] = hm[1, 1, 40, 3] = 9.0
    hm[2, 2, 63, 63] = 11.0
    lab = np.maximum(peaky_heatmaps(402, 3, 21, 64, 64).numpy(), 0)
    return hm, lab


def test_g4_argmax_accuracy_bit_exact():
    g = golden('g4_argmax_accuracy')
    hm, lab = _g4_inputs()
    preds, maxvals = ol.get_max_preds(hm)
    assert np.array_equal(preds, g['preds']) and np.array_equal(maxvals, g['maxvals'])
    assert tuple(preds[1, 0]) == (0.0, 0.0) and tuple(preds[1, 1]) == (7.0, 10.0)
    acc, avg, cnt, pred = ol.accuracy(hm, lab)
    assert np.array_equal(acc, g['acc']) and avg == float(g['avg']) and cnt == int(g['cnt'])
    assert np.array_equal(pred, g['pred'])


def test_g5_softargmax():
    hm = randn(501, 2, 21, 64, 64, scale=0.05)
    hm[0, 0, 20, 33] += 1.0
    np.testing.assert_allclose(ol.soft_argmax(hm).numpy(), golden('g5_softargmax')['uv'], **T)


def test_g6_gl_schedule():
    g = golden('g6_gl')
    for i, lam in zip(g['iters'], g['lam']):
        assert abs(op.gl_coeff(int(i)) - lam) < 1e-8  # golden lam went through an fp32 grad (eps 6e-9 at 0.1)
        layer = op.WarmStartGradientLayer(1.0, 0.0, 0.1, 1000, False)
        layer.iter_num = int(i)
        x = torch.ones(4, requires_grad=True)
        layer(x).sum().backward()
        assert float(x.grad[0]) == lam
    assert op.gl_coeff(0) == 0.0 and abs(op.gl_coeff(1000) - 0.0462117) < 1e-6


def test_g7_full_iteration():
    g = golden('g7_iteration')
    bb = make_backbone('resnet18')
    model = op.PoseResNetx9(bb, op.Upsampling(bb.out_features), 256, 21)
    fill_module_(model, 701)
    B = 2
    x_s, x_t = randn(702, B, 3, 256, 256), randn(703, B, 3, 256, 256)
    label_s = rand(704, B, 21, 64, 64) * (rand(705, B, 21, 64, 64) > 0.9)
    w_s, w_t = weights_bk(706, B, 21), weights_bk(707, B, 21)
    model.gl_layer.iter_num = 500
    tr = DATrainer(model)
    for it in range(2):
        out = tr.step(x_s, label_s, w_s, x_t, w_t)
        got = np.array([float(out['loss_s']), float(out['loss_gf']), float(out['loss_gt'])])
        np.testing.assert_allclose(got, g[f'it{it}_losses'], rtol=2e-5)
        if it == 0:
            np.testing.assert_allclose(out['y_s'][:, ::5].numpy(), g['it0_y_s'], rtol=1e-4, atol=1e-5)
    sd = model.state_dict()
    keys = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    assert keys == list(g['param_keys'])
    s = np.array([float(sd[k].double().sum()) for k in keys])
    a = np.array([float(sd[k].double().abs().sum()) for k in keys])
    np.testing.assert_allclose(a, g['param_abs'], rtol=1e-5)
    np.testing.assert_allclose(s, g['param_sum'], rtol=1e-4, atol=1e-4)


def test_g8_bottleneck_iteration_and_resnet101_forward():
    """Bottleneck nets (ResNet-50 full A/B/C iteration, ResNet-101 forward) against the values captured from the
    reference's PoseResNetx9 / loss classes (make_golden.py:g8_bottleneck)."""
    g = golden('g8_bottleneck')
    bb = make_backbone('resnet50')
    model = op.PoseResNetx9(bb, op.Upsampling(bb.out_features), 256, 21)
    fill_module_(model, 801)
    B = 2
    x_s, x_t = randn(802, B, 3, 256, 256), randn(8034, B, 3, 256, 256)
    label_s = rand(804, B, 21, 64, 64) * (rand(805, B, 21, 64, 64) > 0.9)
    w_s, w_t = weights_bk(806, B, 21), weights_bk(807, B, 21)
    model.gl_layer.iter_num = 500
    out = DATrainer(model).step(x_s, label_s, w_s, x_t, w_t)
    got = np.array([float(out['loss_s']), float(out['loss_gf']), float(out['loss_gt'])])
    np.testing.assert_allclose(got, g['losses'], rtol=2e-5)
    np.testing.assert_allclose(out['y_s'][:, ::5].numpy(), g['y_s'], rtol=1e-4, atol=1e-5)
    sd = model.state_dict()
    keys = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    assert keys == list(g['param_keys'])
    a = np.array([float(sd[k].double().abs().sum()) for k in keys])
    np.testing.assert_allclose(a, g['param_abs'], rtol=1e-5)
    assert int(sd['backbone.layer4.2.bn3.num_batches_tracked']) == int(g['nbt_layer4']) == 3
    bb = make_backbone('resnet101')
    m = op.PoseResNetx9(bb, op.Upsampling(bb.out_features), 256, 21)
    fill_module_(m, 811)
    m.train()
    x = randn(812, B, 3, 256, 256)
    with torch.no_grad():
        y, _, _, y_adv3, f = m(x)
    np.testing.assert_allclose(y[:, ::5].numpy(), g['r101_y'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y_adv3[:, ::5].numpy(), g['r101_y_adv3'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.state_dict()['backbone.layer3.22.bn3.running_mean'].numpy(), g['r101_rm'], rtol=1e-4, atol=1e-6)


def test_synthetic_labels_follow_generate_target():
    """The package's label generator (used for synthetic batches) against the oracle's restatement of
    uda/dataset/util.py:9-68, including centres on / outside the border."""
    from utils.synthetic import generate_target as gt_pkg
    rng = np.random.default_rng(3)
    joints = rng.uniform(-10, 266, size=(21, 2))
    joints[0] = (0.0, 0.0); joints[1] = (255.9, 255.9); joints[2] = (3.9, 250.0); joints[3] = (258.0, 10.0)
    vis = (rng.random((21, 1)) < 0.8).astype(np.float32)
    a, wa = ol.generate_target(joints, vis, (64, 64), 2, (256, 256))
    b, wb = gt_pkg(joints, vis, (64, 64), 2, (256, 256))
    assert np.array_equal(a, b) and np.array_equal(wa, wb)


def test_g9_generate_target_matches_reference():
    """uda/dataset/util.py:9-68 captured from the reference itself (make_golden.py:g9_generate_target): the oracle's
    restatement and the package's host-side generator, bit-exact, at 256/128/512 images (64/32/128 heat-maps), with
    centres on and across every border, far outside the map, and invisible joints."""
    from seeded import g9_inputs
    from utils.synthetic import generate_target as gt_pkg
    g = golden('g9_generate_target')
    kp, vis = g9_inputs()
    for tag, (hm, img) in dict(a=((64, 64), (256, 256)), b=((32, 32), (128, 128)), c=((128, 128), (512, 512))).items():
        for fn in (ol.generate_target, gt_pkg):
            for b in range(kp.shape[0]):
                t, w = fn(kp[b] * (img[0] / 256.0), vis[b], hm, 2, img)
                assert np.array_equal(t, g['target_' + tag][b]) and np.array_equal(w, g['weight_' + tag][b]), (tag, b, fn.__module__)
    # int() truncates toward zero: x = -2.1 px -> -0.025 -> column 0 (kept); far outside -> weight 0
    assert g['weight_a'][0, 5, 0] == 1 and g['weight_a'][0, 4, 0] == 1 and g['weight_a'][1, 6, 0] == 0 and g['weight_a'][1, 7, 0] == 0

