"""Drop-in API behaviour on the GPU: the reference's call signatures, return types and error behaviour
(uda.model / utils mirrors), checked against the oracle / golden vectors."""
import numpy as np
import pytest
import torch

from conftest import golden
from seeded import fill_module_, randn, rand, peaky_heatmaps, weights_bk

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _f32_mode():
    import mi355
    mi355.set_compute_dtype('f32')
    yield
    mi355.set_compute_dtype('bf16')


def test_numpy_decode_api_matches_golden(gpu):
    """get_max_preds / accuracy keep the reference's numpy-in / numpy-out contract (utils/keypoint_detection.py:7-92)."""
    from utils.keypoint_detection import get_max_preds, accuracy
    g = golden('g4_argmax_accuracy')
    hm = peaky_heatmaps(401, 3, 21, 64, 64).numpy()
    hm[1, 0] = 0.5
    hm[1, 1, 10, 7] = hm[1, 1, 40, 3] = 9.0
    hm[2, 2, 63, 63] = 11.0
    lab = np.maximum(peaky_heatmaps(402, 3, 21, 64, 64).numpy(), 0)
    preds, maxvals = get_max_preds(hm)
    assert isinstance(preds, np.ndarray) and preds.dtype == np.float32
    assert np.array_equal(preds, g['preds']) and np.array_equal(maxvals, g['maxvals'])
    acc, avg, cnt, pred = accuracy(hm, lab)
    assert np.array_equal(acc, g['acc']) and avg == float(g['avg']) and cnt == int(g['cnt']) and np.array_equal(pred, g['pred'])
    # torch CUDA tensors are accepted too (no host round trip of the maps)
    acc2, avg2, cnt2, _ = accuracy(torch.from_numpy(hm).to(gpu), torch.from_numpy(lab).to(gpu))
    assert np.array_equal(acc2, acc) and avg2 == avg and cnt2 == cnt
    with pytest.raises(AssertionError):
        get_max_preds(hm[0])            # 3-d input: same assertion as the reference


def test_loss_api_and_errors(gpu):
    from uda.model.loss import JointsKLLoss
    from uda.model.regda_4 import PseudoLabelGenerator
    from uda.model.regda_7 import RegressionDisparityx6, PseudoLabelGenerator01, RegressionDisparityx1
    from oracle import losses as ol
    B, K = 2, 21
    y, y_adv = peaky_heatmaps(201, B, K, 64, 64), randn(202, B, K, 64, 64)
    w = weights_bk(207, B, K)
    rd = RegressionDisparityx6(PseudoLabelGenerator(K, 64, 64), JointsKLLoss(epsilon=1e-7))
    with pytest.raises(AssertionError):
        rd(y.to(gpu), y_adv.to(gpu), None, w.to(gpu), mode='sideways')     # regda_7.py:3610
    # reduction='none' -> (B,) like the reference
    lab = rand(205, B, K, 64, 64) * (rand(206, B, K, 64, 64) > 0.9)
    got = JointsKLLoss(reduction='none')(y_adv.to(gpu), lab.to(gpu), w.to(gpu))
    ref = ol.JointsKLLoss(reduction='none')(y_adv, lab, w)
    assert tuple(got.shape) == (B,) and torch.allclose(got.cpu(), ref, rtol=1e-4)
    # generator returns (ground_truth, ground_false) tensors on the prediction's device
    gt, gf = PseudoLabelGenerator(K, 64, 64)(y.to(gpu))
    gt_ref, gf_ref = ol.PseudoLabelGenerator(K, 64, 64)(y)
    assert gt.device.type == 'cuda' and torch.equal(gt.cpu(), gt_ref) and torch.allclose(gf.cpu(), gf_ref, atol=2e-7)
    rd1 = RegressionDisparityx1(PseudoLabelGenerator01(K), JointsKLLoss(epsilon=1e-7))
    v = rd1(y.to(gpu), randn(204, B, K, 16, 16).to(gpu), w.to(gpu), mode='min')
    assert v.dim() == 0 and rd1.ground_truth.shape == (B, K, 16, 16)


def test_pose_resnet_pretrain_model_and_parameter_groups(gpu):
    """PoseResNet (source-only pre-training, pose_resnet2.py:157-189) forward/backward vs the oracle, get_parameters()
    groups, eval-mode return type of PoseResNetx9, GL step bookkeeping."""
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling, PoseResNet
    from uda.model.regda_7 import PoseResNetx9, PoseResNetx10
    from uda.model.loss import JointsKLLoss
    from oracle.backbone import make_backbone
    from oracle import pose as op, losses as ol
    bb = models.resnet18()
    m = PoseResNet(bb, Upsampling(bb.out_features), 256, 21, True)
    rb = make_backbone('resnet18')
    r = op.PoseResNet(rb, op.Upsampling(rb.out_features), 256, 21, True)
    fill_module_(r, 31)
    m.load_state_dict(r.state_dict())
    m = m.to(gpu)
    x = randn(32, 2, 3, 128, 128)
    lab = rand(33, 2, 21, 32, 32) * (rand(34, 2, 21, 32, 32) > 0.8)
    m.train(); r.train()
    y = m(x.to(gpu)); y_ref = r(x)
    assert float((y.detach().cpu() - y_ref.detach()).abs().max()) <= 1e-3 * float(y_ref.abs().max())
    loss = JointsKLLoss()(y, lab.to(gpu)); loss.backward()
    loss_ref = ol.JointsKLLoss()(y_ref, lab); loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) <= 1e-4 * abs(float(loss_ref))
    g, g_ref = m.head.weight.grad.cpu(), r.head.weight.grad
    assert float((g - g_ref).norm() / g_ref.norm()) < 2e-2
    groups = m.get_parameters(lr=0.5)
    assert [gr['lr'] for gr in groups] == [0.05, 0.5, 0.5] and len(list(groups[0]['params'])) == len(list(bb.parameters()))
    # DA model: 6 groups, eval returns a single tensor, x10 always the 5-tuple, step() advances lambda
    bb2 = models.resnet18()
    da = PoseResNetx9(bb2, Upsampling(bb2.out_features), 256, 21).to(gpu)
    assert len(da.get_parameters(1.0)) == 6 and da.get_parameters(1.0)[0]['lr'] == 0.1
    da.eval()
    with torch.no_grad():
        out = da(x.to(gpu))
    assert torch.is_tensor(out) and tuple(out.shape) == (2, 21, 32, 32)
    da10 = PoseResNetx10(bb2, da.upsampling, 256, 21).to(gpu).eval()
    with torch.no_grad():
        out10 = da10(x.to(gpu))
    assert isinstance(out10, tuple) and len(out10) == 5 and tuple(out10[3].shape) == (2, 21, 8, 8) and tuple(out10[4].shape) == (2, 256, 32, 32)
    lam0 = da.gl_layer.coeff
    da.step()
    assert da.gl_layer.iter_num == 1 and da.gl_layer.coeff > lam0 == 0.0


def test_batch_of_one_and_frozen_parameters(gpu):
    """B=1 (BatchNorm over a single image) and requires_grad=False parameters (the frozen EMA copy, train1.py:115-117)."""
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx10
    bb = models.resnet18()
    m = PoseResNetx10(bb, Upsampling(bb.out_features), 256, 21).to(gpu)
    for p in m.parameters():
        p.requires_grad = False
    m.train()
    x = randn(35, 1, 3, 256, 256).to(gpu)
    y, y_adv, y_adv2, y_adv3, f = m(x)
    assert not y.requires_grad and all(p.grad is None for p in m.parameters())   # nothing to differentiate, nothing written
    assert torch.isfinite(y).all() and tuple(y_adv2.shape) == (1, 21, 32, 32)
    # un-freeze one head: only its parameters receive gradients, the frozen rest stays untouched
    for p in m.head_adv3.parameters():
        p.requires_grad = True
    y, y_adv, y_adv2, y_adv3, f = m(x)
    y_adv3.sum().backward()
    got = {n for n, p in m.named_parameters() if p.grad is not None}
    assert got and all(n.startswith('head_adv3.') for n in got)


def test_device_prefetcher_stages_batches_in_order(gpu):
    """Pinned double-buffered host->HBM staging (utils.data.DevicePrefetcher): every batch arrives intact, in order, on the
    device, for pinned and pageable host tensors alike; non-tensor items pass through; the source's end ends the iterator."""
    from utils.data import DevicePrefetcher, ForeverDataIterator
    torch.manual_seed(0)
    host = [(torch.randn(4, 3, 32, 32), torch.rand(4, 21, 8, 8).pin_memory(), torch.ones(4, 21, 1) * i, {'i': i}) for i in range(7)]
    got = list(DevicePrefetcher(iter(host), gpu))
    assert len(got) == 7
    for (x, l, w, meta), (gx, gl, gw, gmeta) in zip(host, got):
        assert gx.is_cuda and gl.is_cuda and gw.is_cuda and gmeta is meta
        assert torch.equal(gx.cpu(), x) and torch.equal(gl.cpu(), l) and torch.equal(gw.cpu(), w)
    # with the endless iterator of the training loop: batches keep coming, the staging slots are reused
    it = DevicePrefetcher(ForeverDataIterator(host[:3]), gpu)
    seq = [int(next(it)[2][0, 0, 0]) for _ in range(8)]
    assert seq == [0, 1, 2, 0, 1, 2, 0, 1]
    with pytest.raises(ValueError):
        DevicePrefetcher(iter(host), 'cpu')


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_gradient_layer_scales_whatever_consumes_it(gpu, dt):
    """utils/gl.py:8-18: backward = grad * coeff for EVERY consumer of the layer's output.  The mi355 convs fold lambda
    into their dgrad epilogue (no kernel); a consumer that hides the output behind another op (clone, arithmetic, a
    foreign module) must still see the scaled gradient: direct conv, clone -> conv, mixed fan-out and a plain torch
    consumer all give lambda * (unscaled gradient)."""
    import mi355
    from mi355.nn import Conv2d
    from utils.gl import WarmStartGradientLayer
    mi355.set_compute_dtype(dt)
    tdt = mi355.compute_dtype()
    tol = 1e-5 if dt == 'f32' else 2e-2
    conv = Conv2d(16, 16, 3, 1, 1, bias=False).to(gpu)
    conv2 = Conv2d(16, 32, 1, 1, 0, bias=True).to(gpu)
    fill_module_(conv, 41); fill_module_(conv2, 42)
    gl = WarmStartGradientLayer(alpha=1.0, lo=0.0, hi=0.1, max_iters=1000, auto_step=False)
    gl.iter_num = 700
    lam = gl.coeff
    assert 0.03 < lam < 0.04
    x0 = randn(43, 2, 16, 12, 12).to(gpu).to(tdt).contiguous(memory_format=torch.channels_last)
    wsum = randn(44, 2, 16, 12, 12).to(gpu)

    def grad_of(fn):
        x = x0.clone().requires_grad_(True)
        fn(x).backward()
        return x.grad.float()

    loss = lambda y: (y.float() * wsum[:, :y.shape[1]].expand_as(y) if y.shape[1] <= 16 else y.float()).sum()
    base1 = grad_of(lambda x: loss(conv(x)))                       # no gradient layer
    base2 = grad_of(lambda x: loss(conv(x)) + loss(conv2(x)))
    scale = float(base2.abs().max())
    direct = grad_of(lambda x: loss(conv(gl(x))))                  # folded into the dgrad epilogue
    cloned = grad_of(lambda x: loss(conv(gl(x).clone())))          # tag lost: the layer's own backward scales
    mixed = grad_of(lambda x: (lambda f: loss(conv(f)) + loss(conv2(f * 1.0)))(gl(x)))   # one claims, one does not
    plain = grad_of(lambda x: (gl(x).float() * wsum).sum())        # pure torch consumer
    for name, got, ref in (('direct', direct, lam * base1), ('clone', cloned, lam * base1), ('mixed', mixed, lam * base2),
                           ('plain', plain, lam * wsum)):
        err = float((got - ref).abs().max())
        assert err <= tol * lam * max(scale, 1.0), '%s: %.3e' % (name, err)
    # forward is an alias (no copy), like the tagged path it replaces
    x = x0.clone().requires_grad_(True)
    assert gl(x).data_ptr() == x.data_ptr()


def test_fused_sgd_is_a_drop_in_for_torch_sgd(gpu):
    """torch.optim.SGD semantics in the corner cases the A/B/C loop never hits: a parameter that receives no gradient
    between zero_grad() and step() is skipped (torch: grad None), a parameter whose first gradient arrives after the
    flat buffers were laid out is picked up, and gradients accumulated by autograd itself (torch-native module) work."""
    from mi355.nn import Conv2d
    from mi355.optim import FusedSGD
    import mi355.nn as mnn

    def build(opt_cls):
        torch.manual_seed(0)
        a = Conv2d(8, 8, 3, 1, 1, bias=True).to(gpu)
        b = Conv2d(8, 8, 3, 1, 1, bias=True).to(gpu)
        c = Conv2d(8, 8, 1, 1, 0, bias=True).to(gpu)
        lin = torch.nn.Linear(8, 4).to(gpu)                         # torch-native: AccumulateGrad writes its gradient
        for i, m in enumerate((a, b, c, lin)):
            fill_module_(m, 70 + i)
        ps = [p for m in (a, b, c, lin) for p in m.parameters()]
        return (a, b, c, lin), opt_cls(ps, lr=0.05, momentum=0.9, weight_decay=1e-2, nesterov=True)

    def run(mods, opt, schedule):
        a, b, c, lin = mods
        x = randn(75, 2, 8, 6, 6).to(gpu)
        for use_b, use_c in schedule:
            opt.zero_grad()
            y = a(x)
            if use_b:
                y = y + b(x)
            if use_c:
                y = y + c(x)
            out = lin(y.float().mean(dim=(2, 3)))
            (out ** 2).sum().backward()
            opt.step()
        return [p.detach().float().cpu().clone() for m in mods for p in m.parameters()]

    # reference semantics from torch itself, on plain torch modules with the same weights
    def torch_ref(schedule):
        mods = []
        for i, (k, pad) in enumerate(((3, 1), (3, 1), (1, 0))):
            m = torch.nn.Conv2d(8, 8, k, 1, pad).to(gpu)
            fill_module_(m, 70 + i)
            mods.append(m)
        lin = torch.nn.Linear(8, 4).to(gpu)
        fill_module_(lin, 73)
        mods.append(lin)
        opt = torch.optim.SGD([p for m in mods for p in m.parameters()], lr=0.05, momentum=0.9, weight_decay=1e-2, nesterov=True)
        return run(mods, opt, schedule)

    # b skips iteration 2 (after having momentum), c receives its first gradient only in iteration 3
    schedule = [(True, False), (True, False), (False, False), (True, True), (True, True)]
    mods, opt = build(FusedSGD)
    got = run(mods, opt, schedule)
    ref = torch_ref(schedule)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert torch.allclose(g, r, rtol=2e-4, atol=2e-6), 'parameter %d: max diff %.3e' % (i, float((g - r).abs().max()))


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_gradient_fan_in_is_summed_inside_the_dgrad_epilogues(gpu, dt):
    """A tensor consumed by several mi355 convs (the neck output feeds four heads): with a GradFanIn attached the
    consumers accumulate their input gradients into one buffer; the result equals autograd's own sum."""
    import mi355
    from mi355.nn import Conv2d, GradFanIn
    from utils.gl import WarmStartGradientLayer
    mi355.set_compute_dtype(dt)
    tdt = mi355.compute_dtype()
    convs = [Conv2d(16, 16, 3, 1, 1, bias=True).to(gpu), Conv2d(16, 32, 1, 1, 0, bias=False).to(gpu), Conv2d(16, 16, 3, 2, 1, bias=True).to(gpu)]
    for i, c in enumerate(convs):
        fill_module_(c, 90 + i)
    gl = WarmStartGradientLayer(alpha=1.0, lo=0.0, hi=0.1, max_iters=1000, auto_step=False)
    gl.iter_num = 900
    x0 = randn(95, 2, 16, 12, 12).to(gpu).to(tdt).contiguous(memory_format=torch.channels_last)

    def run(fan):
        x = x0.clone().requires_grad_(True)
        h = x * 1.0                                     # a non-leaf, like the neck output
        h = h.contiguous(memory_format=torch.channels_last)
        if fan:
            h._mi_fan = GradFanIn()
        ha = gl(h)                                      # two of the three consumers sit behind the gradient layer
        loss = convs[0](h).float().sum() + 2.0 * convs[1](ha).float().sum() + 3.0 * convs[2](ha).float().pow(2).sum()
        for c in convs:
            for p in c.parameters():
                p.grad = None
        loss.backward()
        return x.grad.float(), [p.grad.clone() for c in convs for p in c.parameters()]

    g_ref, w_ref = run(False)
    g_fan, w_fan = run(True)
    scale = float(g_ref.abs().max())
    assert float((g_fan - g_ref).abs().max()) <= (1e-5 if dt == 'f32' else 2e-2) * scale
    for a, b in zip(w_fan, w_ref):
        assert torch.equal(a, b)


def test_bias_gradient_of_a_conv_in_front_of_a_training_mode_batchnorm(gpu):
    """conv(+bias) -> BatchNorm2d (training) -> ReLU: the bias gradient is the column sum of the BatchNorm's input gradient,
    which is zero per channel up to rounding; the library writes exact zeros instead of reducing (mi355/nn.py _bias_grad).
    The torch reference shows the size of what is dropped; a conv whose output does NOT go into a BatchNorm keeps its real
    bias gradient; the other gradients are unaffected."""
    import torch.nn as tnn
    import mi355
    from mi355.nn import Conv2d, BatchNorm2d, FusedSequential, ReLU
    mi355.set_compute_dtype('f32')
    seq = FusedSequential(Conv2d(16, 32, 3, 1, 1, bias=True), BatchNorm2d(32), ReLU(), Conv2d(32, 16, 1, 1, 0, bias=True)).to(gpu)
    ref = tnn.Sequential(tnn.Conv2d(16, 32, 3, 1, 1, bias=True), tnn.BatchNorm2d(32), tnn.ReLU(), tnn.Conv2d(32, 16, 1, 1, 0, bias=True)).to(gpu)
    fill_module_(seq, 300)
    with torch.no_grad():
        ref[0].weight.copy_(seq[0].weight); ref[0].bias.copy_(seq[0].bias)
        ref[1].weight.copy_(seq[1].weight); ref[1].bias.copy_(seq[1].bias)
        ref[3].weight.copy_(seq[3].weight); ref[3].bias.copy_(seq[3].bias)
    x = randn(301, 4, 16, 10, 10).to(gpu)
    t = randn(302, 4, 16, 10, 10).to(gpu)
    seq.train(); ref.train()
    (seq(x.contiguous(memory_format=torch.channels_last)).float() * t).sum().backward()
    (ref(x) * t).sum().backward()
    assert float(seq[0].bias.grad.abs().max()) == 0.0
    assert float(ref[0].bias.grad.abs().max()) <= 1e-4 * float(ref[3].bias.grad.abs().max())       # rounding noise in the reference
    close = lambda a, b: float((a - b).abs().max()) <= 1e-3 * (float(b.abs().max()) + 1e-6)
    assert close(seq[3].bias.grad, ref[3].bias.grad)
    assert close(seq[0].weight.grad, ref[0].weight.grad) and close(seq[3].weight.grad, ref[3].weight.grad)
    assert close(seq[1].weight.grad, ref[1].weight.grad) and close(seq[1].bias.grad, ref[1].bias.grad)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_eval_forward_folds_batchnorm_and_replays_from_a_graph(gpu, dt):
    """test.py's forward-only path: (1) in eval mode under no_grad every conv -> BatchNorm (-> + identity) -> ReLU runs as one
    launch with the running statistics folded into the conv's operands -- same heat-maps as the unfolded kernels within
    rounding; (2) GraphedForward replays that forward from a HIP graph, follows parameter / statistic changes (a training
    iteration in between) and leaves other shapes to the eager path."""
    import mi355
    import mi355.nn as mnn
    from mi355.infer import GraphedForward
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling, PoseResNet
    mi355.set_compute_dtype(dt)
    try:
        torch.manual_seed(3)
        bb = models.resnet50(pretrained=False)
        model = PoseResNet(bb, Upsampling(bb.out_features), 256, 21, finetune=True).to(gpu)
        x = randn(400, 4, 3, 64, 64).to(gpu)
        model.train()
        for _ in range(2):                                   # running statistics away from their initial values
            model(x)
        model.eval()
        tol = 3e-2 if dt == 'bf16' else 1e-4
        with torch.no_grad():
            y_fold = model(x)
            mnn._EVAL_FOLD = False
            try:
                y_plain = model(x)
            finally:
                mnn._EVAL_FOLD = True
            scale = float(y_plain.abs().max())
            assert float((y_fold - y_plain).abs().max()) <= tol * scale
            fwd = GraphedForward(model, warmup=1)
            outs = [fwd(x) for _ in range(4)]                # eager, capture + replay, replay, replay
            for o in outs:
                assert torch.equal(o, y_fold)
            assert len(fwd._graphs) == 1
            y_small = fwd(x[:2])                             # another shape: eager until seen often enough
            assert float((y_small - model(x[:2])).abs().max()) == 0.0
        model.train(); model(x); model.eval()                # statistics moved: the graphs must go
        with torch.no_grad():
            y_new = model(x)
            assert torch.equal(fwd(x), y_new) and not torch.equal(y_new, y_fold)
    finally:
        mi355.set_compute_dtype('f32')


@pytest.mark.parametrize('case', [(2, 128, 16, 16, 128, 3, 1, 32), (3, 256, 8, 8, 256, 3, 2, 32), (2, 64, 12, 12, 128, 3, 1, 4)])
def test_grouped_conv_layer_matches_torch(gpu, case):
    """mi355.nn.Conv2d(groups = g) -- the 3x3 conv of a ResNeXt bottleneck (reference resnet.py:124-149 via torchvision's
    `groups`) -- against F.conv2d(groups = g) in fp32: output, input gradient, weight gradient in the grouped (Co, Ci/g, 3, 3) layout
    (first backward overwrites, second accumulates), eval-mode forward with the BatchNorm that follows folded in."""
    import torch.nn.functional as F
    from mi355 import nn as mnn
    N, Ci, H, W, Co, k, s, G = case
    conv = mnn.Conv2d(Ci, Co, k, s, 1, bias=False, groups=G).to(gpu)
    assert tuple(conv.weight.shape) == (Co, Ci // G, k, k)
    w = randn(91, Co, Ci // G, k, k, scale=1.0 / np.sqrt(Ci // G * k * k))
    with torch.no_grad():
        conv.weight.copy_(w.to(gpu))
    x = randn(92, N, Ci, H, W)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride=s, padding=1, groups=G)
    dy = randn(93, *y_ref.shape)
    y_ref.backward(dy)
    from mi355 import ops
    xd = ops.to_nhwc(x.to(gpu), torch.float32).detach().requires_grad_(True)      # a feature map as the layers exchange them (channels_last)
    y = conv(xd)
    assert float((y.detach().float().cpu() - y_ref.detach()).abs().max()) <= 1e-3 * float(y_ref.detach().abs().max())
    y.backward(dy.to(gpu))
    assert float((xd.grad.float().cpu() - xr.grad).abs().max()) <= 1e-3 * float(xr.grad.abs().max())
    assert tuple(conv.weight.grad.shape) == (Co, Ci // G, k, k)
    assert float((conv.weight.grad.cpu() - wr.grad).abs().max()) <= 1e-3 * float(wr.grad.abs().max())
    conv(xd).backward(dy.to(gpu))                        # second backward: accumulates
    assert float((conv.weight.grad.cpu() - 2 * wr.grad).abs().max()) <= 2e-3 * float(wr.grad.abs().max())
    # inference: conv -> eval-mode BatchNorm -> ReLU as one launch on the dense block-diagonal weights
    bn = mnn.BatchNorm2d(Co).to(gpu)
    with torch.no_grad():
        bn.running_mean.copy_(randn(94, Co, scale=0.1).to(gpu)); bn.running_var.copy_((1 + 0.2 * rand(95, Co)).to(gpu))
        bn.weight.copy_((1 + 0.1 * randn(96, Co)).to(gpu)); bn.bias.copy_((0.1 * randn(97, Co)).to(gpu))
    seq = mnn.FusedSequential(conv, bn, mnn.ReLU()).eval()
    mnn.link_conv_bn(seq)
    with torch.no_grad():
        z = seq(x.to(gpu))
        z_ref = F.relu(F.batch_norm(y_ref.detach(), bn.running_mean.cpu(), bn.running_var.cpu(), bn.weight.cpu(), bn.bias.cpu(), False, 0.0, bn.eps))
    assert float((z.float().cpu() - z_ref).abs().max()) <= 1e-3 * float(z_ref.abs().max())


@pytest.mark.parametrize('arch', ['resnext50_32x4d', 'wide_resnet50_2'])
def test_resnext_and_wide_resnet_backbones_train_through_the_pose_model(gpu, arch):
    """The constructors the reference re-exports beside resnet50 / resnet101 (resnet.py:124-183) plug into PoseResNetx9 like those:
    out_features 2048, one step-A forward + backward leaves a finite gradient on every backbone conv (grouped ones in their
    grouped layout), and the eval forward agrees with the train-mode graph's shapes."""
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    bb = models.__dict__[arch](pretrained=False)
    assert bb.out_features == 2048
    model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    fill_module_(model, 901)
    model = model.to(gpu).train()
    x = randn(902, 2, 3, 128, 128).to(gpu)
    y, y_adv, y_adv2, y_adv3, f = model(x)
    assert tuple(y.shape) == (2, 21, 32, 32) and tuple(f.shape)[1] == 256
    (y.square().mean() + y_adv.square().mean()).backward()
    for name, p in model.backbone.named_parameters():
        if name.startswith('fc.'):
            continue
        assert p.grad is not None and tuple(p.grad.shape) == tuple(p.shape) and bool(torch.isfinite(p.grad).all()), name
    if arch.startswith('resnext'):
        assert tuple(model.backbone.layer2[1].conv2.weight.shape) == (256, 8, 3, 3)
        assert float(model.backbone.layer2[1].conv2.weight.grad.abs().sum()) > 0
    model.eval()
    with torch.no_grad():
        ye = model(x)
    ye = ye[0] if isinstance(ye, (tuple, list)) else ye
    assert tuple(ye.shape) == (2, 21, 32, 32) and bool(torch.isfinite(ye).all())


@pytest.mark.parametrize('hw', [(32, 48), (31, 33)])
def test_stem_layer_folded_and_padded_forms_match_torch(gpu, hw):
    """mi355.nn.Conv2d(3, 64, 7, 2, 3): even extents take the folded 4x4 form (space-to-depth image), odd extents the 7x7 form over the
    channel-padded image; both against F.conv2d in fp32 -- output and the (64, 3, 7, 7) weight gradient (overwrite, then accumulate)."""
    import torch.nn.functional as F
    from mi355 import nn as mnn
    H, W = hw
    conv = mnn.Conv2d(3, 64, 7, 2, 3, bias=False).to(gpu)
    assert conv._s2d_ok(torch.empty(2, 3, H, W)) == (H % 2 == 0 and W % 2 == 0)
    w = randn(111, 64, 3, 7, 7, scale=1.0 / np.sqrt(147))
    with torch.no_grad():
        conv.weight.copy_(w.to(gpu))
    x = randn(112, 2, 3, H, W)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(x, wr, None, stride=2, padding=3)
    dy = randn(113, *y_ref.shape)
    y_ref.backward(dy)
    y = conv(x.to(gpu))
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert float((y.detach().float().cpu() - y_ref.detach()).abs().max()) <= 1e-4 * float(y_ref.detach().abs().max())
    y.backward(dy.to(gpu))
    assert float((conv.weight.grad.cpu() - wr.grad).abs().max()) <= 1e-4 * float(wr.grad.abs().max())
    conv(x.to(gpu)).backward(dy.to(gpu))
    assert float((conv.weight.grad.cpu() - 2 * wr.grad).abs().max()) <= 2e-4 * float(wr.grad.abs().max())
