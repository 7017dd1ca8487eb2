"""End-to-end parity on the GPU against golden vectors captured from the reference's own classes
(tests/golden/make_golden.py) and against the CPU oracle, through the drop-in uda.model API.
fp32 mode: north-star tolerance 1e-3 (of the tensor's scale); bf16 mode: stated per test."""
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden
from seeded import fill_module_, randn, rand, peaky_heatmaps, weights_bk

pytestmark = pytest.mark.gpu


class _Feat(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.out_features = c

    def forward(self, x):
        return x


@pytest.fixture(autouse=True)
def _f32_mode():
    import mi355
    mi355.set_compute_dtype('f32')
    yield
    mi355.set_compute_dtype('bf16')


def _close(got, ref, tol=1e-3, name=''):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = np.asarray(ref)
    scale = float(np.abs(ref).max()) + 1e-12
    err = float(np.abs(got - ref).max())
    assert err <= tol * scale, '%s: max err %.3e vs scale %.3e' % (name, err, scale)


def test_g1_neck_and_heads_match_reference(gpu):
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    g = golden('g1_neck_heads')
    m = PoseResNetx9(_Feat(64), Upsampling(64), 256, 21)
    fill_module_(m, 101)
    m = m.to(gpu)
    x = randn(102, 2, 64, 8, 8).to(gpu)
    m.train()
    y, y_adv, y_adv2, y_adv3, f = m(x)
    assert tuple(f.shape) == (2, 256, 64, 64) and y.dtype == torch.float32 and y.is_contiguous()
    for name, t in dict(y=y, y_adv=y_adv, y_adv2=y_adv2, y_adv3=y_adv3).items():
        _close(t, g[name], name=name)
    _close(f[:, :4, :8, :8], g['f_slice'], name='f')
    sd = m.state_dict()
    for k in ('upsampling.1.running_mean', 'upsampling.1.running_var', 'head_adv3.last_lay.6.running_mean',
              'head_adv3.last_lay.6.running_var'):
        _close(sd[k], g[k.replace('.', '_')], name=k)
    assert int(sd['upsampling.1.num_batches_tracked']) == 1
    m.eval()
    with torch.no_grad():
        _close(m(x), g['y_eval'], name='y_eval')
    # bf16 throughput path: same network within bf16 rounding through ~10 layers
    import mi355
    mi355.set_compute_dtype('bf16')
    with torch.no_grad():
        _close(m(x), g['y_eval'], tol=4e-2, name='y_eval_bf16')


def test_g2_losses_and_gradients_match_reference(gpu):
    from mi355 import ops
    from uda.model.loss import JointsKLLoss
    from uda.model.regda_4 import PseudoLabelGenerator
    from uda.model.regda_7 import (PseudoLabelGenerator01, PseudoLabelGenerator03, RegressionDisparityx1,
                                   RegressionDisparityx5, RegressionDisparityx6)
    g = golden('g2_losses')
    B, K = 2, 21
    y = peaky_heatmaps(201, B, K, 64, 64).to(gpu)
    y_adv, y_adv2, y_adv3 = (randn(s, B, K, r, r).to(gpu) for s, r in ((202, 64), (203, 32), (204, 16)))
    label = (rand(205, B, K, 64, 64) * (rand(206, B, K, 64, 64) > 0.9)).to(gpu)
    w = weights_bk(207, B, K).to(gpu)
    target5 = ops.bilinear_up(y_adv2, 64, 1.0, out=ops.bilinear_up(y_adv3, 64, 0.5))
    target0 = ops.bilinear_up(y_adv3, 32)
    kl0, kl7 = JointsKLLoss(), JointsKLLoss(epsilon=1e-7)
    rd6 = RegressionDisparityx6(PseudoLabelGenerator(K, 64, 64), JointsKLLoss(epsilon=1e-7))
    rd5 = RegressionDisparityx5(PseudoLabelGenerator03(K), JointsKLLoss(epsilon=1e-7))
    rd1 = RegressionDisparityx1(PseudoLabelGenerator01(K), JointsKLLoss(epsilon=1e-7))
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
        p = inp.clone().requires_grad_(True)
        v = 3.0 * fn(p)                 # the scalar factor travels through autograd to the device-side backward
        v.backward()
        assert abs(float(v) / 3.0 - float(g[name])) <= 1e-4 * abs(float(g[name])), name
        _close(p.grad[:, ::5] / 3.0, g[name + '_grad'], tol=1e-3, name=name + '_grad')


def test_state_dict_layout_and_checkpoint_interop(gpu):
    """Same keys/shapes as the oracle (= torchvision + reference layout); a checkpoint written by this
    implementation loads into the plain-torch model and vice versa, giving the same eval output."""
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from oracle.train_step import build_model
    ref = build_model('resnet18')
    bb = models.__dict__['resnet18'](pretrained=False)
    mine = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    sd_r, sd_m = ref.state_dict(), mine.state_dict()
    assert list(sd_r.keys()) == list(sd_m.keys()) and len(sd_m) == 222
    assert all(tuple(sd_r[k].shape) == tuple(sd_m[k].shape) for k in sd_r)
    fill_module_(ref, 55)
    buf = io.BytesIO()
    torch.save({'model': ref.state_dict()}, buf)
    buf.seek(0)
    mine.load_state_dict(torch.load(buf)['model'])
    mine = mine.to(gpu)
    x = randn(56, 2, 3, 128, 128)
    ref.eval(); mine.eval()
    with torch.no_grad():
        y_ref = ref(x)
        y = mine(x.to(gpu))
    _close(y, y_ref.numpy(), name='eval y (128x128 input -> 32x32 heat-maps)')
    buf = io.BytesIO()
    torch.save({'model': mine.state_dict()}, buf)
    buf.seek(0)
    ref2 = build_model('resnet18')
    ref2.load_state_dict(torch.load(buf)['model'])
    ref2.eval()
    with torch.no_grad():
        assert torch.allclose(ref2(x), y_ref, atol=1e-6)
    import uda.model as models2
    assert len(PoseResNetx9(models2.resnet50(), Upsampling(2048), 256, 21).state_dict()) == 420


def _g7_setup(gpu):
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from mi355.da_step import build_training
    bb = models.resnet18(pretrained=False)
    model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    fill_module_(model, 701)
    model = model.to(gpu)
    B = 2
    batch = dict(x_s=randn(702, B, 3, 256, 256), x_t=randn(703, B, 3, 256, 256),
                 label_s=rand(704, B, 21, 64, 64) * (rand(705, B, 21, 64, 64) > 0.9),
                 w_s=weights_bk(706, B, 21), w_t=weights_bk(707, B, 21))
    batch = {k: v.to(gpu) for k, v in batch.items()}
    model.gl_layer.iter_num = 500
    step, opts, scheds = build_training(model)
    return model, step, opts, scheds, batch


def _check_g7(model, g):
    sd = model.state_dict()
    keys = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    assert keys == list(g['param_keys'])
    a = np.array([float(sd[k].double().abs().sum()) for k in keys])
    rel = np.abs(a - g['param_abs']) / (np.abs(g['param_abs']) + 1e-12)
    assert rel.max() <= 1e-3, 'param |sum| mismatch: %s rel %.3e' % (keys[int(rel.argmax())], rel.max())


@pytest.mark.parametrize('skip', [True, False])
def test_g7_full_iteration_matches_reference(gpu, skip):
    """Two complete A/B/C iterations (ResNet-18 layout, 256x256, B=2, fp32) against values produced by the
    reference's own model / loss classes: the three losses, a slice of y_s, and every parameter's |sum|."""
    g = golden('g7_iteration')
    model, step, opts, scheds, batch = _g7_setup(gpu)
    step.skip = skip
    for it in range(2):
        out = step.run(batch)
        for s in scheds.values():
            s.step()
        got = np.array([float(out['loss_s']), float(out['loss_gf']), float(out['loss_gt'])])
        # iteration 0 runs on identical weights: 1e-3.  Iteration 1 runs on weights that already differ in the last
        # fp32 bits; the pseudo-labels are arg-max based, and one near-tie (top-2 margin 4e-5 in map (0,13) of this
        # fixture) flips between any two fp32 implementations, moving loss_s by ~1e-3: 5e-3 there.
        np.testing.assert_allclose(got, g[f'it{it}_losses'], rtol=1e-3 if it == 0 else 5e-3)
        if it == 0:
            _close(out['y_s'][:, ::5], g['it0_y_s'], name='y_s')
    _check_g7(model, g)
    assert model.backbone.fc.weight.grad is None      # never touched, like torch.optim.SGD with grad None


def test_g7_graph_replay_matches_eager(gpu):
    """The captured-graph iteration is the same computation: 4 eager iterations, capture, 2 replays, compared
    with a twin trained eagerly for 6 iterations (LR schedule and GL lambda advance through device scalars)."""
    m1, s1, o1, sch1, batch = _g7_setup(gpu)
    m2, s2, o2, sch2, _ = _g7_setup(gpu)
    for _ in range(6):
        s2.run(batch)
        for s in sch2.values():
            s.step()
    for _ in range(4):
        s1.run(batch)
        for s in sch1.values():
            s.step()
    s1.capture(batch, warmup=0)
    for _ in range(2):
        s1.replay(batch)
        for s in sch1.values():
            s.step()
    torch.cuda.synchronize()
    assert m1.gl_layer.iter_num == m2.gl_layer.iter_num == 506
    for (k, a), b in zip(m1.state_dict().items(), m2.state_dict().values()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, b, rtol=2e-4, atol=1e-6), k
        else:
            assert torch.equal(a, b), k


def test_gradients_within_the_reference_fp32_noise(gpu):
    """Whole-network gradient parity.  ReLU masks and BatchNorm projections make the gradient of this network
    ill-conditioned: the CPU oracle itself moves by ~1e-2 (relative L2, per parameter) between fp32 and fp64.
    So the HIP fp32 path is held to that yardstick: its distance to the fp64 oracle must be within 3x the fp32
    oracle's own distance to it (+1e-4), for every parameter of step A's backward."""
    from oracle.backbone import make_backbone
    from oracle import pose as op
    from oracle.train_step import DATrainer
    model, step, opts, scheds, batch = _g7_setup(gpu)
    step._fwdbwd_A(batch)
    mine = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
    G = {}
    for dt in (torch.float32, torch.float64):
        bb = make_backbone('resnet18')
        ref = op.PoseResNetx9(bb, op.Upsampling(bb.out_features), 256, 21)
        fill_module_(ref, 701)
        ref.gl_layer.iter_num = 500
        ref = ref.to(dt).train()
        tr = DATrainer(ref)
        b = {k: v.cpu().to(dt) for k, v in batch.items()}
        y_s, y_s_adv, y_s_adv2, y_s_adv3, _ = ref(b['x_s'])
        l = 2 * tr.criterion(y_s, b['label_s'], b['w_s']) + 4 * tr.rd2(y_s, y_s_adv2, None, b['w_s'], mode='min') + \
            4 * tr.rd(y_s, y_s_adv, None, b['w_s'], mode='min') + 4 * tr.rd1(y_s, y_s_adv3, b['w_s'], mode='min')
        l.backward()
        G[dt] = {k: p.grad.double() for k, p in ref.named_parameters() if p.grad is not None}
    assert set(mine) == set(G[torch.float64])
    worst = 0.0
    for k, t in G[torch.float64].items():
        if float(t.abs().max()) < 1e-6:
            continue                      # biases in front of a BatchNorm: exact gradient is 0, only noise remains
        n = float(t.norm())
        e_ref = float((G[torch.float32][k] - t).norm()) / n
        e_mine = float((mine[k] - t).norm()) / n
        worst = max(worst, e_mine)
        assert e_mine <= 3 * e_ref + 1e-4, '%s: mine %.3e vs fp32 oracle %.3e' % (k, e_mine, e_ref)
    assert worst < 5e-2


def _supervised_grads(gpu, dtype_name, bnbwd=False):
    """Gradients of the supervised loss only (main head + neck + backbone).  The adversarial losses of step A take their
    targets from an arg-max of the prediction: on this random-init fixture any rounding change flips arg-max positions
    and with them the targets, so those gradients are not comparable across precisions."""
    import mi355
    import mi355.nn as mnn
    from uda.model.loss import JointsKLLoss
    mi355.set_compute_dtype(dtype_name)
    old = mnn._FUSE_BNBWD
    mnn._FUSE_BNBWD = bnbwd
    try:
        model, step, opts, scheds, batch = _g7_setup(gpu)
        model.train()
        y = model(batch['x_s'])[0]
        JointsKLLoss()(y, batch['label_s'], batch['w_s']).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
    finally:
        mnn._FUSE_BNBWD = old
        mi355.set_compute_dtype('f32')


def test_bf16_gradients_track_the_fp32_path(gpu):
    """Whole-network gradients of the throughput mode (bf16 storage / MFMA bf16, fp32 accumulate).  This random-init fixture
    amplifies rounding by ~2.5e5 from head to stem (fp32 vs fp64 oracle: 1.5e-2), so no bf16 implementation can be close
    to fp32 deep in the network; the yardstick is torch's own bfloat16 CPU run of the oracle model on the same fixture:
    for every parameter the HIP bf16 path must be as close to the fp64 oracle as that run is (2x its distance + 3e-2;
    observed: ours 0.02 .. 0.57, torch CPU bf16 0.02 .. 0.6), main-head parameters within 0.2, and every gradient must
    agree in direction with the fp32 path (cosine >= 0.6).  Layer-exact bf16 evidence: tests/test_gpu_bf16_layers.py."""
    from oracle.backbone import make_backbone
    from oracle import pose as op
    from oracle import losses as ol
    g32 = _supervised_grads(gpu, 'f32')
    g16 = _supervised_grads(gpu, 'bf16')
    assert set(g32) == set(g16)
    _, _, _, _, batch = _g7_setup(gpu)
    G = {}
    for dt in (torch.float64, torch.bfloat16):
        bb = make_backbone('resnet18')
        ref = op.PoseResNetx9(bb, op.Upsampling(bb.out_features), 256, 21)
        fill_module_(ref, 701)
        ref = ref.to(dt).train()
        y = ref(batch['x_s'].cpu().to(dt))[0]
        acc = torch.float64 if dt == torch.float64 else torch.float32
        ol.JointsKLLoss()(y.to(acc), batch['label_s'].cpu().to(acc), batch['w_s'].cpu().to(acc)).backward()
        G[dt] = {k: p.grad.double() for k, p in ref.named_parameters() if p.grad is not None}
    assert set(G[torch.float64]) == set(g16)
    for k, t in g32.items():
        if float(t.abs().max()) < 1e-6:
            continue
        e = float((g16[k] - t).norm()) / float(t.norm())
        cos = float((g16[k] * t).sum()) / (float(g16[k].norm()) * float(t.norm()) + 1e-30)
        assert torch.isfinite(g16[k]).all(), k
        assert cos >= 0.6, '%s: cosine %.3f' % (k, cos)
        if k.startswith('head.'):
            assert e <= 0.2, '%s: bf16 vs fp32 relative L2 %.3e' % (k, e)
        t64 = G[torch.float64][k]
        n64 = float(t64.norm())
        e_mine = float((g16[k] - t64).norm()) / n64
        e_torch = float((G[torch.bfloat16][k] - t64).norm()) / n64
        assert e_mine <= 2 * e_torch + 3e-2, '%s: HIP bf16 %.3e from fp64, torch CPU bf16 %.3e' % (k, e_mine, e_torch)


def test_optin_bn_backward_fusion_matches_default(gpu):
    """MI355_BN_BWD_FUSE path (BatchNorm-backward reduction inside the dgrad epilogue): same gradients as the default
    path up to summation order (fp32 mode: 1e-4 relative L2)."""
    a = _supervised_grads(gpu, 'f32', bnbwd=False)
    b = _supervised_grads(gpu, 'f32', bnbwd=True)
    assert set(a) == set(b)
    for k, t in a.items():
        if float(t.abs().max()) < 1e-6:
            continue
        e = float((b[k] - t).norm()) / float(t.norm())
        assert e <= 1e-4, '%s: %.3e' % (k, e)


def test_bf16_training_reduces_the_supervised_loss(gpu):
    """End-to-end sanity of the throughput mode: 80 A/B/C iterations on one fixed synthetic batch (ResNet-18, 128x128, B=4,
    bf16, guarded normalisation as in bench.py) stay finite and bring the supervised KL loss down by more than a third."""
    import mi355
    import uda.model as models
    from mi355.da_step import build_training
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from utils.synthetic import make_batch
    mi355.set_compute_dtype('bf16')
    try:
        torch.manual_seed(1)
        bb = models.resnet18(pretrained=False)
        model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(gpu)
        step, opts, scheds = build_training(model, heatmap_size=32)
        for c in step.crit.values():
            if hasattr(c, 'guard_empty_maps'):
                c.guard_empty_maps = True
        batch = make_batch(4, 128, 32, seed=3, device=gpu)
        first = last = None
        for it in range(80):
            out = step.run(batch)
            for s in scheds.values():
                s.step()
            if it == 0:
                first = float(out['loss_s'])
        last = [float(out[k]) for k in ('loss_s', 'loss_gf', 'loss_gt')]
        assert all(v == v for v in last), last
        assert last[0] < 0.66 * first, (first, last)
    finally:
        mi355.set_compute_dtype('f32')


# ---------------------------------------------------------------- Bottleneck nets (the benchmarked architectures)
def _g8_setup(gpu, arch='resnet50', seed=801):
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    bb = models.__dict__[arch](pretrained=False)
    model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    fill_module_(model, seed)
    return model.to(gpu)


def _g8_batch(gpu):
    B = 2
    batch = dict(x_s=randn(802, B, 3, 256, 256), x_t=randn(8034, B, 3, 256, 256),
                 label_s=rand(804, B, 21, 64, 64) * (rand(805, B, 21, 64, 64) > 0.9),
                 w_s=weights_bk(806, B, 21), w_t=weights_bk(807, B, 21))
    return {k: v.to(gpu) for k, v in batch.items()}


@pytest.mark.parametrize('skip', [True, False])
def test_g8_resnet50_iteration_matches_reference(gpu, skip):
    """BASELINE config 2's network (ResNet-50: Bottleneck blocks, uda/model/resnet.py:92-107; 1x1 conv1 with the fused
    residual-fork gradient, stride-2 3x3 beside a stride-2 1x1 downsample, 2048-channel BatchNorm): one complete A/B/C
    iteration (train1.py:371-458, B=2, 256x256, fp32 mode) against values produced by the reference's own
    PoseResNetx9 / loss classes (make_golden.py:g8_bottleneck): step-A forward 5-tuple, per-parameter gradient norms of
    step A, the three losses, every parameter's |sum| after the iteration."""
    from mi355.da_step import build_training
    g = golden('g8_bottleneck')
    model = _g8_setup(gpu)
    batch = _g8_batch(gpu)
    model.gl_layer.iter_num = 500
    assert len(model.state_dict()) == 420
    step, opts, scheds = build_training(model)
    step.skip = skip
    # step-A forward + backward alone first (gradients are consumed by the update inside run())
    step._fwdbwd_A(batch)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
    assert sorted(grads) == list(g['gradA_keys'])
    # Yardstick for everything that depends on gradients: the reference's own classes run in fp64 (golden keys *64).
    # Gradients of this random-init network are ill-conditioned (reference fp32 vs fp64: up to 7e-3 on a norm), so the
    # HIP fp32 path must be as close to fp64 as the reference's fp32 is (3x its distance + 1e-3).  Gradients that are
    # exactly 0 in exact arithmetic (biases in front of a BatchNorm) hold rounding noise only and are skipped.
    # Per parameter the bound is 3x the reference's own fp32 distance for that parameter + 0.75x the distance of the reference's
    # WORST parameter (5.3e-3): a parameter on which the reference's fp32 run happened to land close to fp64 (4e-4) is not
    # thereby better conditioned -- two equally exact stem kernels (7x7 over the padded image / 4x4 over the folded one, both
    # 6e-7 of the output scale from fp64: profiles/stem_s2d_err.py) move single parameters between 0.2x and 1.1x of a bound built
    # on their own reference distance + 3e-3, while median (6.1e-4 / 5.9e-4, reference 5.3e-4) and maximum (8.9e-3 / 6.8e-3,
    # reference 7.1e-3) of the distribution do not move (profiles/g8_grad_deviation.py).
    n32, n64 = g['gradA_norm'], g['gradA_norm64']
    keep = [(k, a32, a64) for k, a32, a64 in zip(g['gradA_keys'], n32, n64) if a64 >= 1e-6 * n64.max()]
    ref_rel = [abs(a32 - a64) / a64 for _, a32, a64 in keep]
    floor = max(3e-3, 0.75 * max(ref_rel))
    mine_rel = []
    for k, a32, a64 in keep:
        n = float(grads[str(k)].norm())
        mine_rel.append(abs(n - a64) / a64)
        assert abs(n - a64) <= 3 * abs(a32 - a64) + floor * a64, 'step-A gradient norm of %s: %.6e vs fp64 %.6e (ref fp32 %.6e)' % (k, n, a64, a32)
    assert np.median(mine_rel) <= 3 * np.median(ref_rel) + 1e-4, (np.median(mine_rel), np.median(ref_rel))
    assert max(mine_rel) <= 3 * max(ref_rel), (max(mine_rel), max(ref_rel))
    model2 = _g8_setup(gpu)
    model2.train()
    y_s, y_s_adv, y_s_adv2, y_s_adv3, f_s = model2(batch['x_s'])
    for name, t in dict(y_s=y_s, y_s_adv=y_s_adv, y_s_adv2=y_s_adv2, y_s_adv3=y_s_adv3).items():
        _close(t[:, ::5], g[name], name=name)
    _close(f_s[:, :8, :16, :16], g['f_slice'], name='f')
    assert abs(float(f_s.detach().double().abs().sum()) - float(g['f_abs'])) <= 1e-3 * float(g['f_abs'])
    # that forward updated the BN running statistics once: start again from the seeded state
    fill_module_(model2, 801)
    model2.gl_layer.iter_num = 500
    step, opts, scheds = build_training(model2)
    step.skip = skip
    out = step.run(batch)
    for s in scheds.values():
        s.step()
    got = np.array([float(out['loss_s']), float(out['loss_gf']), float(out['loss_gt'])])
    assert abs(got[0] - g['losses'][0]) <= 1e-3 * g['losses'][0]          # step A: identical weights
    # Steps B, C run on the weights step A updated with those ill-conditioned gradients: fp64 yardstick (reference fp32 vs fp64:
    # 1.1e-3 / 5.8e-4).  Forward kernels that are equally exact but round differently land anywhere within ~5e-3 of fp64 on
    # loss_gf -- 7x7 stem + fused statistics 6.4e-4, folded stem 4.6e-3, 7x7 stem + stand-alone statistics pass 5.4e-3
    # (profiles/r04_stem_s2d.txt) -- so the bound is 3x the reference's distance + 5e-3, not + 1e-3.
    for i in (1, 2):
        l32, l64 = g['losses'][i], g['losses64'][i]
        assert abs(got[i] - l64) <= 3 * abs(l32 - l64) + 5e-3 * abs(l64), (i, got[i], l32, l64)
    sd = model2.state_dict()
    keys = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    assert keys == list(g['param_keys'])
    a = np.array([float(sd[k].double().abs().sum()) for k in keys])
    rel = np.abs(a - g['param_abs']) / (np.abs(g['param_abs']) + 1e-12)
    assert rel.max() <= 1e-3, 'param |sum| mismatch: %s rel %.3e' % (keys[int(rel.argmax())], rel.max())
    assert int(sd['backbone.layer4.2.bn3.num_batches_tracked']) == int(g['nbt_layer4']) == 3
    assert model2.backbone.fc.weight.grad is None


def test_g8_resnet101_forward_matches_reference(gpu):
    """BASELINE config 3's network (ResNet-101, 23 Bottleneck blocks in layer3): train-mode forward, BN running
    statistics of the deepest stages (fp64 yardstick, see make_golden.py:g8_bottleneck), eval-mode forward 1e-3, bf16 eval
    heat-maps 6e-2 (through 100+ layers)."""
    import mi355
    g = golden('g8_bottleneck')
    m = _g8_setup(gpu, 'resnet101', 811)
    assert len(m.state_dict()) == 726
    x = randn(812, 2, 3, 256, 256).to(gpu)
    m.train()
    with torch.no_grad():
        y, y_adv, y_adv2, y_adv3, f = m(x)
    def close64(t, key):
        """as close to the reference run in fp64 as the reference's own fp32 run is (3x its distance + 1e-3 of the scale)"""
        r32, r64 = np.asarray(g[key], np.float64), np.asarray(g[key + '64'])
        scale = float(np.abs(r64).max())
        gap = float(np.abs(r32 - r64).max())
        err = float(np.abs(t.detach().double().cpu().numpy() - r64).max())
        assert err <= 3 * gap + 1e-3 * scale, '%s: %.3e from fp64 (reference fp32: %.3e, scale %.3e)' % (key, err, gap, scale)
    close64(y[:, ::5], 'r101_y')
    close64(y_adv3[:, ::5], 'r101_y_adv3')
    close64(f[:, :8, :16, :16], 'r101_f_slice')
    assert abs(float(f.double().abs().sum()) - float(g['r101_f_abs64'])) <= 1e-3 * float(g['r101_f_abs64'])
    sd = m.state_dict()
    close64(sd['backbone.layer3.22.bn3.running_mean'], 'r101_rm')
    close64(sd['backbone.layer4.2.bn3.running_var'], 'r101_rv')
    m.eval()
    with torch.no_grad():
        _close(m(x)[:, ::5], g['r101_y_eval'], name='y_eval')
        mi355.set_compute_dtype('bf16')
        _close(m(x)[:, ::5], g['r101_y_eval'], tol=6e-2, name='y_eval_bf16')


@pytest.mark.parametrize('arch,last_bn', [('resnet50', 'backbone.layer4.2.bn3'), ('resnet101', 'backbone.layer3.22.bn3')])
def test_full_size_bf16_iteration_properties(gpu, arch, last_bn):
    """BASELINE configs 2 and 3 at full size (ResNet-50 / ResNet-101, 256x256, B=64 source + 64 target, bf16): no oracle finishes this in
    seconds, so size-independent properties: finite losses, the shared B/C part of the network updates its running
    statistics twice (num_batches_tracked: A once + B/C twice = 3 per iteration; adversarial heads: 3 forwards),
    backbone.fc never receives a gradient, every stepped parameter moved and stays finite over two iterations."""
    import mi355
    import uda.model as models
    from mi355.da_step import build_training
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from utils.synthetic import make_batch
    mi355.set_compute_dtype('bf16')
    try:
        torch.manual_seed(1)
        bb = models.__dict__[arch](pretrained=False)
        model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(gpu)
        before = {k: v.detach().clone() for k, v in model.named_parameters()}
        step, opts, scheds = build_training(model)
        batch = make_batch(64, 256, 64, seed=1, device=gpu)
        out = step.run(batch)
        for s in scheds.values():
            s.step()
        torch.cuda.synchronize()
        vals = [float(out[k]) for k in ('loss_s', 'loss_gf', 'loss_gt')]
        assert all(np.isfinite(v) for v in vals), vals
        sd = model.state_dict()
        assert int(sd[last_bn + '.num_batches_tracked']) == 3
        assert int(sd['upsampling.7.num_batches_tracked']) == 3
        assert int(sd['head.1.num_batches_tracked']) == 3
        assert int(sd['head_adv3.last_lay.6.num_batches_tracked']) == 3
        assert model.backbone.fc.weight.grad is None and model.backbone.fc.bias.grad is None
        for k, p in model.named_parameters():
            if k.startswith('backbone.fc.'):
                assert torch.equal(p, before[k]), k
            else:
                assert torch.isfinite(p).all(), k
                # a conv bias in front of a training-mode BatchNorm starts at 0 and its gradient is exactly 0 (the column sum
                # of a BatchNorm input gradient; the reference moves it by rounding noise): it may stay where it was
                inert = k.endswith('.bias') and p.grad is not None and float(p.grad.abs().max()) == 0.0 and float(before[k].abs().max()) == 0.0
                assert inert or not torch.equal(p, before[k]), 'parameter %s was not updated' % k
        out2 = step.run(batch)
        torch.cuda.synchronize()
        assert all(np.isfinite(float(out2[k])) for k in ('loss_s', 'loss_gf', 'loss_gt'))
    finally:
        mi355.set_compute_dtype('f32')
