"""Per-kernel parity on the GPU: every C-ABI entry point (called through mi355.ops, i.e. ctypes) against a
plain torch fp32 CPU computation of the same op / the CPU oracle.  fp32 mode: 1e-3 of the output scale
(north-star tolerance); bf16 mode: inputs pre-rounded to bf16, tolerance = bf16 output rounding (2^-8 rel)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from seeded import randn, rand, peaky_heatmaps, weights_bk

pytestmark = pytest.mark.gpu

DT = {'f32': torch.float32, 'bf16': torch.bfloat16}


def _ops():
    import mi355
    from mi355 import ops
    mi355.load()
    return ops


def _tol(dt, ref):
    scale = float(ref.abs().max()) + 1e-12
    return (1e-3 if dt == 'f32' else 1.2e-2) * scale


def _round(t, dt):
    return t.to(DT[dt]).float()


def _nhwc(t_nchw, dt, dev, cpad=None):
    """CPU NCHW fp32 -> device channels_last tensor of dtype dt (through the library's own converter)."""
    ops = _ops()
    return ops.to_nhwc(t_nchw.to(dev), DT[dt], cpad)


def _back(t):
    return t.float().cpu().contiguous()


CONV_CASES = [
    # N, Ci, H, W, Co, k, s, p
    (2, 64, 16, 16, 64, 3, 1, 1),
    (2, 64, 16, 16, 128, 3, 2, 1),
    (3, 128, 8, 8, 256, 1, 1, 0),
    (2, 64, 16, 16, 256, 1, 2, 0),
    (2, 3, 32, 32, 64, 7, 2, 3),      # stem: Ci padded to one 16-byte chunk
    (2, 256, 16, 16, 64, 4, 2, 1),    # conv-form of ConvTranspose2d(64 -> 256, 4, 2, 1)
    (1, 64, 9, 13, 64, 3, 1, 1),      # ragged spatial size (M not a tile multiple)
    (5, 128, 12, 12, 64, 3, 2, 1),    # odd batch, M = 180 rows (partial tiles)
    # 3x3 / stride 1 with power-of-two widths: the kw-shared wgrad kernel (bf16), one case per LDS segment size
    (3, 64, 8, 8, 128, 3, 1, 1),      # W = 8: eight image rows per 64-pixel tile, M = 192
    (1, 128, 32, 32, 256, 3, 1, 1),   # W = 32, two output-channel tiles x two input-channel tiles
    (1, 64, 20, 64, 128, 3, 1, 1),    # W = 64: one image row per tile, H not a power of two
    (1, 64, 6, 128, 64, 3, 1, 1),     # W = 128: half rows with halo pixels fetched from the neighbours
    # 3x3 / 4x4 stride 2: the parity-image wgrad kernel (bf16), one case per output width
    (1, 64, 24, 32, 128, 3, 2, 1),    # Wo = 16, Ho = 12 (not a power of two)
    (1, 64, 64, 64, 64, 4, 2, 1),     # 4x4 (conv-form of a transposed conv), Wo = 32, 64 output channels
    (1, 64, 128, 128, 128, 3, 2, 1),  # Wo = 64: one image row per tile
    (2, 128, 32, 32, 256, 4, 2, 1),   # 4x4, two output-channel tiles x two input-channel tiles
    (32, 64, 64, 64, 256, 3, 1, 1),   # 2048 output tiles: the shared-A-tile (KW3) forward / dgrad kernel is chosen by default
    (64, 64, 64, 64, 256, 3, 1, 1),   # 4096 output tiles: its 256x128 macro-tile build
    # variants only the benchmark's sizes select, against torch at full size (round-2 verdict, item 8)
    (64, 256, 64, 64, 256, 3, 2, 1),  # wgrad_kw2 at stage size + 4-phase strided dgrad + LDS-DMA ring forward (3x3 s2 256->256 @64->32, B=64)
    (64, 256, 16, 16, 256, 3, 1, 1),  # 256 tiles, K = 2304: the LDS-DMA ring on the mid-size layers (3x3 256->256 @16x16, B=64)
    (64, 512, 8, 8, 512, 3, 1, 1),    # 128 tiles of 128 x 128: 64-row tiles with two K groups per workgroup (3x3 512->512 @8x8, B=64)
    (64, 1024, 16, 16, 256, 1, 1, 0),  # 256 tiles, K = 1024 in one tap: two K groups on a 1x1 conv (1024->256 @16x16, B=64)
    (64, 64, 64, 64, 64, 3, 1, 1),     # 2048 tiles of 128 x 64: the shared-A-tile (KW3) kernel's build for 64 output channels (layer1's 3x3 convs, B=64)
    (64, 256, 16, 16, 1024, 1, 1, 0),  # 256 tiles of 256 x 256 (forward; short K): the one-tile-per-CU LDS-DMA build where it is dispatched
    (64, 256, 32, 32, 256, 3, 1, 1),   # 256 tiles of 256 x 256, K = 2304 (3x3 256->256 @32x32, B=64)
]


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(gpu, dt, case):
    ops = _ops()
    N, Ci, H, W, Co, k, s, p = case
    per = 8 if dt == 'bf16' else 4
    Cip = ((Ci + per - 1) // per) * per
    x = _round(randn(1, N, Ci, H, W), dt)
    w = _round(randn(2, Co, Ci, k, k, scale=1.0 / np.sqrt(Ci * k * k)), dt)
    bias = randn(3, Co, scale=0.1)
    y_ref = F.conv2d(x, w, bias, stride=s, padding=p)
    dy = _round(randn(4, *y_ref.shape), dt)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=s, padding=p).backward(dy)

    desc = ops.make_desc(N, H, W, Cip, Co, k, k, s, p, DT[dt])
    xd = _nhwc(x, dt, gpu, Cip)
    # master weights in [Co][kh][kw][Ci] memory order
    wm = w.permute(0, 2, 3, 1).contiguous().to(gpu)
    wf, wt = ops.pack_weights(wm, Co, k * k, Ci, Cip, DT[dt])
    y = ops.conv_fwd(desc, xd, wf, bias.to(gpu))
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert float((_back(y) - y_ref).abs().max()) <= _tol(dt, y_ref)

    dyd = _nhwc(dy, dt, gpu)
    dx = ops.conv_dgrad(desc, dyd, wt)
    dx_ref = xr.grad
    assert float((_back(dx)[:, :Ci] - dx_ref).abs().max()) <= _tol(dt, dx_ref)
    # accumulate + device scale (the GL fold): dx2 = dx + 0.25 * dgrad
    sc = torch.tensor(0.25, device=gpu)
    dx2 = ops.conv_dgrad(desc, dyd, wt, scale_dev=sc, out=dx.clone(), accumulate=True)
    assert float((_back(dx2)[:, :Ci] - 1.25 * dx_ref).abs().max()) <= 2 * _tol(dt, dx_ref)

    dw = torch.full((Co, k, k, Cip), 7.0, dtype=torch.float32, device=gpu)   # overwritten when accumulate=0
    ops.conv_wgrad(desc, xd, dyd, dw, accumulate=False)
    dw_ref = wr.grad.permute(0, 2, 3, 1)
    got = dw.cpu()[..., :Ci]
    assert float((got - dw_ref).abs().max()) <= _tol(dt, dw_ref)
    ops.conv_wgrad(desc, xd, dyd, dw, accumulate=True)
    assert float((dw.cpu()[..., :Ci] - 2 * dw_ref).abs().max()) <= 2 * _tol(dt, dw_ref)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 32, 32), (3, 20, 36), (64, 256, 256)])
def test_stem_as_4x4_conv_over_the_space_to_depth_image(gpu, dt, shape):
    """Conv2d(3, 64, 7, stride 2, padding 3) (reference resnet.py:23-28) in its folded form: mi355_nchw_to_s2d + mi355_stem_s2d_pack
    + the cropped 4x4 / unit-stride descriptor + mi355_stem_s2d_unpack_grad, against torch's 7x7 conv (forward, BatchNorm
    statistics of the output, weight gradient with and without accumulation); the last shape is the benchmark's."""
    ops = _ops()
    N, H, W = shape
    Co = 64
    x = _round(randn(61, N, 3, H, W), dt)
    w = _round(randn(62, Co, 3, 7, 7, scale=1.0 / np.sqrt(147)), dt)
    y_ref = F.conv2d(x, w, None, stride=2, padding=3)
    dy = _round(randn(63, *y_ref.shape), dt)
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, stride=2, padding=3).backward(dy)
    dw_ref = wr.grad.permute(0, 2, 3, 1).contiguous()

    xs = ops.to_nhwc_s2d(x.to(gpu), DT[dt])
    assert tuple(xs.shape) == (N, 16, H // 2, W // 2)
    fold = x.view(N, 3, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4)              # [N][dy][dx][c][by][bx]
    ref_s = torch.zeros(N, 2, 2, 4, H // 2, W // 2); ref_s[:, :, :, :3] = fold
    assert torch.equal(_back(xs), ref_s.reshape(N, 16, H // 2, W // 2))
    wm = w.permute(0, 2, 3, 1).contiguous().to(gpu)
    wf = ops.stem_s2d_pack(wm, DT[dt])
    desc = ops.make_desc(N, H // 2, W // 2, 16, Co, 4, 4, 1, 2, DT[dt], out_hw=(H // 2, W // 2))
    y, part = ops.conv_fwd_stats(desc, xs, wf, None)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert float((_back(y) - y_ref).abs().max()) <= _tol(dt, y_ref)
    assert part is not None and part[1] >= 1
    outs = []
    for pp in (None, part):             # BatchNorm from the epilogue's partials == BatchNorm with its own statistics pass
        rm, rv = torch.zeros(Co, device=gpu), torch.ones(Co, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        outs.append(ops.bn_train_fwd(y, None, torch.ones(Co, device=gpu), torch.zeros(Co, device=gpu), rm, rv, nbt, 1e-5, 0.1, True, partial=pp))
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6) and torch.allclose(outs[0][2], outs[1][2], rtol=2e-5, atol=1e-6)

    dyd = _nhwc(dy, dt, gpu)
    gs = torch.empty(Co * 256, dtype=torch.float32, device=gpu)
    ops.conv_wgrad(desc, xs, dyd, gs, accumulate=False)
    g = torch.full((Co, 7, 7, 3), 3.0, dtype=torch.float32, device=gpu)
    ops.stem_s2d_unpack_grad(gs, g, False)
    assert float((g.cpu() - dw_ref).abs().max()) <= _tol(dt, dw_ref)
    ops.stem_s2d_unpack_grad(gs, g, True)
    assert float((g.cpu() - 2 * dw_ref).abs().max()) <= 2 * _tol(dt, dw_ref)
    # the positions of the folded gradient that correspond to no 7x7 tap are exactly the products with the zero channel / the
    # kh = -1 row: they are computed (the image is not zero there) but never read back
    with pytest.raises(Exception):      # a cropped descriptor stays refused where it has no meaning (input gradient)
        ops.conv_dgrad(desc, dyd, wf)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 64, 16, 16, 128, 3, 1, 13, 11), (1, 64, 9, 20, 64, 4, 2, 9, 20), (2, 128, 8, 8, 64, 5, 2, 5, 7)])
def test_cropped_output_descriptor(gpu, dt, case):
    """Ho x Wo smaller than the full output of a unit-stride conv = its top-left crop, in the forward and the weight gradient (the
    3x3 case must NOT take the shifted-row weight-gradient kernel, which needs same-size maps); the input gradient refuses it."""
    ops = _ops()
    N, Ci, H, W, Co, k, p, Ho, Wo = case
    x = _round(randn(71, N, Ci, H, W), dt)
    w = _round(randn(72, Co, Ci, k, k, scale=1.0 / np.sqrt(Ci * k * k)), dt)
    wr = w.clone().requires_grad_(True)
    y_full = F.conv2d(x, wr, None, stride=1, padding=p)
    assert Ho <= y_full.shape[2] and Wo <= y_full.shape[3]
    y_ref = y_full[:, :, :Ho, :Wo]
    dy = _round(randn(73, *y_ref.shape), dt)
    y_ref.backward(dy)
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, 1, p, DT[dt], out_hw=(Ho, Wo))
    xd = _nhwc(x, dt, gpu)
    wf, wt = ops.pack_weights(w.permute(0, 2, 3, 1).contiguous().to(gpu), Co, k * k, Ci, Ci, DT[dt])
    y = ops.conv_fwd(desc, xd, wf, None)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert float((_back(y) - y_ref.detach()).abs().max()) <= _tol(dt, y_ref.detach())
    dw = torch.empty((Co, k, k, Ci), dtype=torch.float32, device=gpu)
    ops.conv_wgrad(desc, xd, _nhwc(dy, dt, gpu), dw, accumulate=False)
    dw_ref = wr.grad.permute(0, 2, 3, 1)
    assert float((dw.cpu() - dw_ref).abs().max()) <= _tol(dt, dw_ref)
    with pytest.raises(Exception):
        ops.conv_dgrad(desc, _nhwc(dy, dt, gpu), wt)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_conv_residual_epilogue(gpu, dt):
    ops = _ops()
    N, Ci, H, W, Co = 2, 64, 8, 8, 128
    x = _round(randn(11, N, Ci, H, W), dt)
    w = _round(randn(12, Co, Ci, 1, 1, scale=0.1), dt)
    r = _round(randn(13, N, Co, H, W), dt)
    ref = F.conv2d(x, w) + r
    desc = ops.make_desc(N, H, W, Ci, Co, 1, 1, 1, 0, DT[dt])
    wf, _ = ops.pack_weights(w.permute(0, 2, 3, 1).contiguous().to(gpu), Co, 1, Ci, Ci, DT[dt])
    y = ops.conv_fwd(desc, _nhwc(x, dt, gpu), wf, None, _nhwc(r, dt, gpu))
    assert float((_back(y) - ref).abs().max()) <= _tol(dt, ref)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 64, 16, 16, 64, 1, 1, 0), (3, 256, 8, 8, 64, 1, 1, 0), (32, 64, 64, 64, 256, 3, 1, 1), (2, 64, 16, 16, 128, 3, 2, 1),
                                  (64, 256, 16, 16, 256, 3, 1, 1)])      # the last one: two K groups per workgroup
def test_conv_dgrad_accumulates_onto_a_bit_masked_gradient(gpu, dt, case):
    """dx <- dgrad + (bit ? dx : 0): the fork of a residual block whose identity-branch gradient still lacks the block's
    ReLU mask (mi355_conv_dgrad_masked_acc), and the stand-alone masking kernel; both against plain tensor arithmetic, and
    run twice (an earlier build of this epilogue dropped addends now and then -- see the comment in csrc/igemm.hip)."""
    ops = _ops()
    N, Ci, H, W, Co, k, s, p = case
    per = 8 if dt == 'bf16' else 4
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, DT[dt])
    g = torch.Generator(device='cpu').manual_seed(77 + Ci + Co)
    wm = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(gpu)
    _, wt = ops.pack_weights(wm, Co, k * k, Ci, Ci, DT[dt])
    dy = ops.nhwc_empty(N, Co, desc.Ho, desc.Wo, DT[dt], gpu).normal_()
    base = ops.nhwc_empty(N, Ci, H, W, DT[dt], gpu).normal_()
    mask = torch.randint(0, 256, (N * H * W * Ci // per,), dtype=torch.uint8, device=gpu)
    bits = ((mask.view(-1, 1) >> torch.arange(per, device=gpu).view(1, per)) & 1).view(N, H, W, Ci).permute(0, 3, 1, 2)
    dx = ops.conv_dgrad(desc, dy, wt)
    ref = (base.float() * bits + dx.float())
    tol = (2e-2 if dt == 'bf16' else 1e-5) * (float(ref.abs().max()) + 1)
    outs = []
    for _ in range(2):
        out = ops.conv_dgrad_masked_acc(desc, dy, wt, base.clone(), mask)
        assert float((out.float() - ref).abs().max()) <= tol
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    masked = ops.apply_relu_mask(base.clone(), mask)
    assert torch.equal(masked.float(), base.float() * bits)
    with pytest.raises(Exception):
        ops.conv_dgrad_masked_acc(desc, dy, wt, base.clone(), None)


@pytest.mark.parametrize('case', [CONV_CASES[0], CONV_CASES[1], CONV_CASES[5], CONV_CASES[9], CONV_CASES[16], CONV_CASES[17]])
def test_conv_kernels_give_the_same_bits_run_after_run(gpu, case):
    """Every epilogue form of the bf16 conv kernels twice on the same operands: forward (+bias, +residual, fused ReLU, fused
    statistics), input gradient (plain, scaled accumulate, masked accumulate) and weight gradient must reproduce bit for bit.
    Guards against timing-dependent faults: an earlier build of the masked accumulate dropped addends in a few lanes now and then
    while every single-shot tolerance test passed (csrc/igemm.hip, epilogue comment)."""
    ops = _ops()
    N, Ci, H, W, Co, k, s, p = case
    dt = torch.bfloat16
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, dt)
    g = torch.Generator(device='cpu').manual_seed(1000 + Ci + Co + k)
    wm = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(gpu)
    wf, wt = ops.pack_weights(wm, Co, k * k, Ci, Ci, dt)
    x = ops.nhwc_empty(N, Ci, H, W, dt, gpu).normal_()
    dy = ops.nhwc_empty(N, Co, desc.Ho, desc.Wo, dt, gpu).normal_()
    res = ops.nhwc_empty(N, Co, desc.Ho, desc.Wo, dt, gpu).normal_()
    base = ops.nhwc_empty(N, Ci, H, W, dt, gpu).normal_()
    bias = torch.randn(Co, generator=g).to(gpu)
    mask = torch.randint(0, 256, (N * H * W * Ci // 8,), dtype=torch.uint8, device=gpu)
    sc = torch.tensor(0.25, device=gpu)
    dw = torch.zeros(Co, k, k, Ci, device=gpu)

    def once():
        out = [ops.conv_fwd(desc, x, wf, bias), ops.conv_fwd(desc, x, wf, bias, res), ops.conv_fwd(desc, x, wf, bias, res, relu=True)]
        y, part = ops.conv_fwd_stats(desc, x, wf, None)
        out += [y, part[0][:part[1] * Co * 3].clone()]
        out += [ops.conv_dgrad(desc, dy, wt), ops.conv_dgrad(desc, dy, wt, scale_dev=sc, out=base.clone(), accumulate=True),
                ops.conv_dgrad_masked_acc(desc, dy, wt, base.clone(), mask)]
        ops.conv_wgrad(desc, x, dy, dw, accumulate=False)
        out.append(dw.clone())
        return out
    a, b = once(), once()
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), 'output %d differs between two runs' % i


PGEMM_CASES = [
    # N, Ci, H, W, Co : 1x1 / unit-stride layers; one case per column-tile / ring choice of dispatch_pgemm, ragged rows, column tails
    (3, 128, 8, 8, 256),        # 192 rows: below the kernel's floor -> stays on the gather kernel whatever the mode (control)
    (2, 1024, 16, 16, 256),     # K = 1024: no weight slice fits LDS -> stays on the gather kernel (control)
    (3, 64, 9, 13, 256),        # 351 rows (ragged: partial row tiles), one K step, two 128-wide column tiles
    (4, 128, 16, 16, 64),       # two K steps, 64-wide column tile
    (16, 64, 64, 64, 64),       # 65536 rows: many row tiles per persistent block, ring of 8
    (8, 256, 32, 32, 256),      # four K steps, 64-KB weight slice, one block per CU
    (8, 512, 16, 16, 128),      # K = 512: 64-wide column tiles, eight K steps
    (5, 128, 24, 24, 256),      # 2880 rows = 45 row tiles over two column tiles
    (64, 256, 16, 16, 1024),    # eight column tiles
    (64, 64, 64, 64, 256),      # make-or-break shape of the iteration: 64 -> 256 @64x64, B = 64
]


@pytest.mark.parametrize('case', PGEMM_CASES)
def test_pgemm_matches_the_gather_kernel_bit_for_bit(gpu, case):
    """The weights-stationary streaming GEMM (csrc/pgemm.hip) and the gather kernel run the same MFMA sequence per output element: every
    epilogue form must give IDENTICAL bits on both (forward plain / bias / bias + residual + ReLU, input gradient plain / scaled
    accumulate / bit-masked accumulate); the fused BatchNorm statistics must agree per channel once folded; and the forward
    must agree with a torch matmul within the bf16 tolerance."""
    import mi355
    ops = _ops()
    lib = mi355.load()
    N, Ci, H, W, Co = case
    dt = torch.bfloat16
    desc = ops.make_desc(N, H, W, Ci, Co, 1, 1, 1, 0, dt)
    g = torch.Generator(device='cpu').manual_seed(31 + Ci + Co)
    wm = (torch.randn(Co, 1, 1, Ci, generator=g) * 0.05).to(gpu)
    wf, wt = ops.pack_weights(wm, Co, 1, Ci, Ci, dt)
    x = ops.nhwc_empty(N, Ci, H, W, dt, gpu).normal_()
    dy = ops.nhwc_empty(N, Co, H, W, dt, gpu).normal_()
    res = ops.nhwc_empty(N, Co, H, W, dt, gpu).normal_()
    base = ops.nhwc_empty(N, Ci, H, W, dt, gpu).normal_()
    bias = torch.randn(Co, generator=g).to(gpu)
    mask = torch.randint(0, 256, (N * H * W * Ci // 8,), dtype=torch.uint8, device=gpu)
    sc = torch.tensor(0.25, device=gpu)
    M = N * H * W

    def run():
        out = [ops.conv_fwd(desc, x, wf), ops.conv_fwd(desc, x, wf, bias), ops.conv_fwd(desc, x, wf, bias, res, relu=True)]
        y, part = ops.conv_fwd_stats(desc, x, wf, None)
        pr = part[0][:part[1] * Co * 3].view(part[1], Co, 3).double()
        n, mean, m2 = pr[..., 0], pr[..., 1], pr[..., 2]
        tot = n.sum(0); gm = (n * mean).sum(0) / tot
        var = (m2 + n * (mean - gm) ** 2).sum(0) / tot
        out += [y, ops.conv_dgrad(desc, dy, wt), ops.conv_dgrad(desc, dy, wt, scale_dev=sc, out=base.clone(), accumulate=True),
                ops.conv_dgrad_masked_acc(desc, dy, wt, base.clone(), mask)]
        torch.cuda.synchronize()
        return out, (tot, gm, var)
    prev = lib.mi355_set_pgemm(0)
    try:
        ref, st_ref = run()
        lib.mi355_set_pgemm(2)
        got, st_got = run()
        got2, _ = run()
    finally:
        lib.mi355_set_pgemm(prev)
    for i, (a, b, c) in enumerate(zip(ref, got, got2)):
        assert torch.equal(a, b), 'output %d: pgemm differs from the gather kernel' % i
        assert torch.equal(b, c), 'output %d: pgemm differs between two runs' % i
    assert torch.equal(st_ref[0], st_got[0]) and float(st_got[0][0]) == M
    assert float((st_ref[1] - st_got[1]).abs().max()) <= 1e-5 * (float(st_ref[1].abs().max()) + 1)
    assert float((st_ref[2] - st_got[2]).abs().max()) <= 1e-4 * float(st_ref[2].abs().max())
    yref = x.permute(0, 2, 3, 1).reshape(M, Ci).float() @ wm.reshape(Co, Ci).to(dt).float().t()
    err = (got[0].permute(0, 2, 3, 1).reshape(M, Co).float() - yref).abs().max()
    assert float(err) <= 1.2e-2 * float(yref.abs().max())


STAT_CASES = [
    # kind, N, Ci, H, W, Co, k, s, p   (kind 'conv': Conv2d; 'deconv': conv-form of ConvTranspose2d(Co -> Ci, 4, 2, 1))
    ('conv', 2, 64, 16, 16, 128, 3, 1, 1),
    ('conv', 3, 128, 8, 8, 256, 1, 1, 0),
    ('conv', 2, 64, 16, 16, 128, 3, 2, 1),
    ('conv', 2, 3, 32, 32, 64, 7, 2, 3),       # stem (padded input channels, small-C kernel variant)
    ('conv', 1, 64, 9, 13, 64, 3, 1, 1),       # ragged: partial row tiles
    ('conv', 6, 64, 40, 40, 128, 1, 1, 0),     # 9600 rows: 128-row tiles
    ('conv', 20, 64, 64, 64, 128, 1, 1, 0),    # 81920 rows: >= 640 slices -> the block-per-channel finalize kernel
    ('deconv', 2, 256, 16, 16, 64, 4, 2, 1),   # four output phases in one launch
    ('conv', 64, 64, 64, 64, 256, 3, 1, 1),    # 4096 output tiles: the 256x128 macro tile of the shared-A-tile kernel with the statistics epilogue
    ('conv', 64, 256, 16, 16, 256, 3, 1, 1),   # 256 output tiles, K = 2304: two K groups per workgroup (intra-workgroup split-K), statistics by all 8 waves
    ('deconv', 64, 256, 8, 8, 256, 4, 2, 1),   # the same through the four phases of a transposed conv (2048 -> 256 @8 -> 16 is the model's; here 256 -> 256)
]


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('case', STAT_CASES)
def test_conv_epilogue_bn_statistics(gpu, dt, case):
    """BatchNorm statistics fused into the conv / deconv epilogue == the stand-alone statistics pass over the same output."""
    ops = _ops()
    kind, N, Ci, H, W, Co, k, s, p = case
    per = 8 if dt == 'bf16' else 4
    Cip = ((Ci + per - 1) // per) * per
    desc = ops.make_desc(N, H, W, Cip, Co, k, k, s, p, DT[dt])
    w = _round(randn(2, Co, Ci, k, k, scale=1.0 / np.sqrt(Ci * k * k)), dt)
    wf, wt = ops.pack_weights(w.permute(0, 2, 3, 1).contiguous().to(gpu), Co, k * k, Ci, Cip, DT[dt])
    if kind == 'conv':
        x = _nhwc(_round(randn(1, N, Ci, H, W) + 0.4, dt), dt, gpu, Cip)
        y, part = ops.conv_fwd_stats(desc, x, wf)
        y0 = ops.conv_fwd(desc, x, wf)
    else:
        x = _nhwc(_round(randn(1, N, Co, desc.Ho, desc.Wo) + 0.4, dt), dt, gpu)
        y, part = ops.conv_dgrad_stats(desc, x, wt)
        y0 = ops.conv_dgrad(desc, x, wt)
    assert part is not None and part[1] >= 1
    assert torch.equal(y, y0)                        # the fused epilogue leaves the conv result untouched
    C = y.shape[1]
    gamma, beta = (1 + 0.1 * randn(23, C)).to(gpu), (0.1 * randn(24, C)).to(gpu)
    outs = []
    for pp in (None, part):
        rm, rv = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        outs.append(ops.bn_train_fwd(y, None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, True, partial=pp) + (rm, rv, nbt))
    (ya, ma, ia, rma, rva, na), (yb, mb, ib, rmb, rvb, nb) = outs
    assert int(na) == int(nb) == 1
    assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-6) and torch.allclose(ia, ib, rtol=2e-5, atol=1e-6)
    assert torch.allclose(rma, rmb, rtol=1e-5, atol=1e-6) and torch.allclose(rva, rvb, rtol=2e-5, atol=1e-6)
    assert float((ya.float() - yb.float()).abs().max()) <= (2e-2 if dt == 'bf16' else 1e-4)   # bf16: one-ulp flips at most


CAT_CASES = [
    # N, Ci, H, W, Co, k, s, p, K : y = conv(x) + conv1x1(heat-map with K channels at the OUTPUT resolution), one case per tile choice
    (2, 256, 16, 16, 256, 1, 1, 0, 21),     # 64 x 64 tiles, register-staged
    (3, 64, 9, 13, 64, 3, 1, 1, 21),        # ragged rows, 64 output channels
    (2, 256, 32, 32, 256, 3, 2, 1, 21),     # strided conv: the heat-map lives at the 16 x 16 output resolution
    (8, 256, 64, 64, 256, 1, 1, 0, 21),     # 64 x 128 tiles (short K, many rows): make_head's shape
    (16, 256, 64, 64, 256, 3, 2, 1, 21),    # LDS-DMA ring (K-heavy, >= 256 tiles): make_head2's shape
    (16, 128, 32, 32, 256, 1, 1, 0, 8),     # 128 x 128 tiles register-staged, one-chunk second operand
    (4, 64, 32, 32, 128, 3, 1, 1, 32),      # 64 x 128 tiles through the row-count rule, a full 32-channel second operand
]


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CAT_CASES)
def test_conv_fwd_concat_k(gpu, dt, case):
    """mi355_conv_fwd_cat: heatmap_conv(hm) + feature_conv(x) (reference regda_7.py:4575, :4651-4654) as one implicit GEMM against
    torch's two convs on the same rounded operands; the fused BatchNorm statistics against the stand-alone statistics pass; two runs
    bit-identical."""
    ops = _ops()
    N, Ci, H, W, Co, k, s, p, K = case
    x = _round(randn(1, N, Ci, H, W), dt)
    w = _round(randn(2, Co, Ci, k, k, scale=1.0 / np.sqrt(Ci * k * k)), dt)
    b1, b2 = randn(3, Co, scale=0.1), randn(4, Co, scale=0.1)
    y1 = F.conv2d(x, w, b1, stride=s, padding=p)
    Ho, Wo = y1.shape[2:]
    hm = _round(randn(5, N, K, Ho, Wo), dt)
    w2 = _round(randn(6, Co, K, 1, 1, scale=0.2), dt)
    ref = y1 + F.conv2d(hm, w2, b2)
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, DT[dt])
    xd = _nhwc(x, dt, gpu)
    wf, _ = ops.pack_weights(w.permute(0, 2, 3, 1).contiguous().to(gpu), Co, k * k, Ci, Ci, DT[dt])
    w2f, w2t = ops.pack_weights(w2.permute(0, 2, 3, 1).contiguous().to(gpu), Co, 1, K, 32, DT[dt])
    hm32 = _nhwc(hm, dt, gpu, 32)
    y = ops.conv_fwd_cat(desc, xd, wf, b1.to(gpu), hm32, w2f.view(Co, 32), b2.to(gpu))
    assert tuple(y.shape) == tuple(ref.shape)
    assert float((_back(y) - ref).abs().max()) <= _tol(dt, ref)
    ys, part = ops.conv_fwd_cat(desc, xd, wf, b1.to(gpu), hm32, w2f.view(Co, 32), b2.to(gpu), want_stats=True)
    assert torch.equal(ys, y) and part is not None and part[1] >= 1
    gamma, beta = (1 + 0.1 * randn(23, Co)).to(gpu), (0.1 * randn(24, Co)).to(gpu)
    outs = []
    for pp in (None, part):
        rm, rv = torch.zeros(Co, device=gpu), torch.ones(Co, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        outs.append(ops.bn_train_fwd(y, None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, True, partial=pp) + (rm, rv))
    (ya, ma, ia, rma, rva), (yb, mb, ib, rmb, rvb) = outs
    assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-6) and torch.allclose(ia, ib, rtol=2e-5, atol=1e-6)
    assert torch.allclose(rma, rmb, rtol=1e-5, atol=1e-6) and torch.allclose(rva, rvb, rtol=2e-5, atol=1e-6)
    # no biases, and the error path: a second operand wider than one K tile is refused
    y0 = ops.conv_fwd_cat(desc, xd, wf, None, hm32, w2f.view(Co, 32), None)
    assert float((_back(y0) - (ref - b1.view(1, -1, 1, 1) - b2.view(1, -1, 1, 1))).abs().max()) <= _tol(dt, ref)
    import mi355
    wide = ops.nhwc_empty(N, 128, Ho, Wo, DT[dt], gpu)
    with pytest.raises(mi355.Mi355Error):
        ops.conv_fwd_cat(desc, xd, wf, None, wide, torch.zeros(Co, 128, dtype=DT[dt], device=gpu), None)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(4, 64, 16, 16), (2, 256, 8, 8), (3, 2048, 4, 4), (2, 24, 5, 7)])
@pytest.mark.parametrize('relu,res', [(True, False), (True, True), (False, False)])
def test_bn_train_fwd_bwd(gpu, dt, shape, relu, res):
    ops = _ops()
    N, C, H, W = shape
    x = _round(randn(21, *shape) * 1.5 + 0.3, dt)
    r = _round(randn(22, *shape), dt) if res else None
    gamma, beta = 1 + 0.1 * randn(23, C), 0.1 * randn(24, C)
    rm, rv = 0.1 * randn(25, C), 1 + 0.2 * rand(26, C)
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if res:
        y_ref = y_ref + rr
    if relu:
        y_ref = F.relu(y_ref)
    dy = _round(randn(27, *shape), dt)
    y_ref.backward(dy)

    rm_d, rv_d = rm.to(gpu), rv.to(gpu)
    nbt = torch.zeros((), dtype=torch.int64, device=gpu)
    xd = _nhwc(x, dt, gpu)
    rd = _nhwc(r, dt, gpu) if res else None
    y, mean, invstd = ops.bn_train_fwd(xd, rd, gamma.to(gpu), beta.to(gpu), rm_d, rv_d, nbt, 1e-5, 0.1, relu)
    tol = _tol(dt, y_ref)
    assert float((_back(y) - y_ref.detach()).abs().max()) <= tol
    assert torch.allclose(rm_d.cpu(), rm_ref, rtol=1e-4, atol=1e-5) and torch.allclose(rv_d.cpu(), rv_ref, rtol=1e-4, atol=1e-5)
    assert int(nbt) == 1
    dg = torch.zeros(C, device=gpu); db = torch.zeros(C, device=gpu)
    # backward consumes the library's own y (bf16-rounded) for the ReLU mask
    # with a residual the ReLU mask needs the forward output y; without it the mask is recomputed from x (y=None)
    dx, dres = ops.bn_bwd(_nhwc(dy, dt, gpu), xd, y if (relu and res) else None, gamma.to(gpu), mean, invstd, dg, db, False, relu, res,
                          beta=beta.to(gpu))
    gtol = _tol(dt, xr.grad) * (3 if dt == 'bf16' else 1)
    assert float((_back(dx) - xr.grad).abs().max()) <= gtol
    if res:
        assert float((_back(dres) - rr.grad).abs().max()) <= _tol(dt, rr.grad)
    assert float((dg.cpu() - gr.grad).abs().max()) <= 2e-3 * (float(gr.grad.abs().max()) + 1) * (8 if dt == 'bf16' else 1)
    assert float((db.cpu() - br.grad).abs().max()) <= 2e-3 * (float(br.grad.abs().max()) + 1) * (8 if dt == 'bf16' else 1)
    if relu and res:
        # the bit mask written by the forward's apply pass replaces y in the backward: same results, bit for bit
        mask = ops.bn_relu_mask(xd)
        y2, _, _ = ops.bn_train_fwd(xd, rd, gamma.to(gpu), beta.to(gpu), rm.to(gpu), rv.to(gpu), nbt.clone(), 1e-5, 0.1, relu, relu_mask=mask)
        assert torch.equal(y2, y)
        per = 8 if dt == 'bf16' else 4
        bits = (mask.view(N * H * W, C // per, 1) >> torch.arange(per, device=gpu).view(1, 1, per)) & 1
        assert torch.equal(bits.view(N, H, W, C).permute(0, 3, 1, 2).bool(), y > 0)
        dg2 = torch.zeros(C, device=gpu); db2 = torch.zeros(C, device=gpu)
        dx2, dres2 = ops.bn_bwd(_nhwc(dy, dt, gpu), xd, None, gamma.to(gpu), mean, invstd, dg2, db2, False, relu, res,
                                beta=beta.to(gpu), relu_mask=mask)
        # (mask-from-y runs as reduce + finalize + apply, the bit-mask form may take the one-launch LDS-resident kernel:
        #  same masked gradient bit for bit, sums folded in another order)
        assert torch.equal(dres2, dres)
        assert float((dx2.float() - dx.float()).abs().max()) <= (2e-2 if dt == 'bf16' else 1e-5) * (float(dx.float().abs().max()) + 1)
        assert torch.allclose(dg2, dg, rtol=1e-4, atol=1e-4) and torch.allclose(db2, db, rtol=1e-4, atol=1e-4)
    # eval mode
    ye = ops.bn_eval_fwd(xd, rd, gamma.to(gpu), beta.to(gpu), rm.to(gpu), rv.to(gpu), 1e-5, relu)
    ye_ref = F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5)
    if res:
        ye_ref = ye_ref + r
    if relu:
        ye_ref = F.relu(ye_ref)
    assert float((_back(ye) - ye_ref).abs().max()) <= _tol(dt, ye_ref)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(64, 256, 16, 16), (64, 2048, 8, 8), (64, 128, 32, 32), (5, 192, 7, 9), (64, 512, 8, 8), (1, 64, 1, 3),
                                   (64, 256, 64, 64)])      # the last one: 134 MB in bf16, only partially LDS-resident
@pytest.mark.parametrize('mode', ['plain', 'relu', 'relu_residual'])
def test_bn_backward_one_launch_lds_resident(gpu, dt, shape, mode):
    """Tensors that fit the chip's LDS take the one-launch backward (csrc/bn.hip bn_bwd_resident_kernel): model-sized and
    ragged shapes, all three mask sources it covers, against torch's fp32 batch_norm backward on the same rounded operands;
    gradient accumulation into dgamma / dbeta; no grid-barrier spin gave up."""
    ops = _ops()
    N, C, H, W = shape
    relu, res = mode != 'plain', mode == 'relu_residual'
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    g = torch.Generator(device='cpu').manual_seed(1234 + C + H)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(tdt).float().to(gpu)
    dy = torch.randn(shape, generator=g).to(tdt).float().to(gpu)
    r = torch.randn(shape, generator=g).to(tdt).float().to(gpu) if res else None
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(gpu); beta = (0.1 * torch.randn(C, generator=g)).to(gpu)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    y_ref = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if res:
        y_ref = y_ref + rr
    if relu:
        y_ref = F.relu(y_ref)
    y_ref.backward(dy)
    to = lambda t: t.to(tdt).contiguous(memory_format=torch.channels_last)
    xd, dyd = to(x), to(dy)
    rm, rv, nbt = torch.zeros(C, device=gpu), torch.ones(C, device=gpu), torch.zeros((), dtype=torch.int64, device=gpu)
    mask = ops.bn_relu_mask(xd) if res else None
    y, mean, invstd = ops.bn_train_fwd(xd, to(r) if res else None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, relu, relu_mask=mask)
    dg = torch.full((C,), 0.5, device=gpu); db = torch.full((C,), -0.25, device=gpu)
    dx, dres = ops.bn_bwd(dyd, xd, None, gamma, mean, invstd, dg, db, True, relu, res, beta=beta, relu_mask=mask)
    # the in-launch exchange folds the blocks' partial sums in a fixed order: a second run reproduces every bit
    dg2 = torch.full((C,), 0.5, device=gpu); db2 = torch.full((C,), -0.25, device=gpu)
    dx2, dres2 = ops.bn_bwd(dyd, xd, None, gamma, mean, invstd, dg2, db2, True, relu, res, beta=beta, relu_mask=mask)
    assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db) and (dres is None or torch.equal(dres2, dres))
    scale = float(xr.grad.abs().max()) + 1e-6
    # fp32: the forward's y = x*sc+sh may round a value next to zero to the other side of the ReLU than torch's formula does
    err = (dx.float() - xr.grad).abs()
    tol = (2e-2 if dt == 'bf16' else 2e-4) * scale
    assert float((err > tol).float().mean()) <= (1e-4 if relu else 0.0), float(err.max())
    if res:
        assert float(((dres.float() - rr.grad).abs() > 1e-6).float().mean()) <= 1e-4
    gt = 4e-3 * (8 if dt == 'bf16' else 1)
    flips = 2 if relu else 0       # a ReLU decision that flips on one element moves that channel's sums by |dy|, |dy * xhat|
    assert int(((dg - 0.5 - gr.grad).abs() > gt * (float(gr.grad.abs().max()) + 1)).sum()) <= flips
    assert int(((db + 0.25 - br.grad).abs() > gt * (float(br.grad.abs().max()) + 1)).sum()) <= flips
    assert ops.bn_resident_timeouts() == 0


def test_bn_backward_one_launch_gives_up_loudly(gpu):
    """A one-launch backward whose blocks cannot exchange their sums must not pass incomplete sums off as gradients: with the
    spin bound shrunk to one poll (test hook) the early arrivers of a multi-block launch give up -> the launch is counted,
    every gradient a failed block wrote is NaN, bn_resident_check raises, switches the one-launch form off and clears the
    counters; after the reset (and the hook restored) the same call is exact again."""
    import mi355
    ops = _ops()
    N, C, H, W = 64, 256, 32, 32                  # 4 channel groups x 64 row blocks: every block has plenty of work before it arrives
    tdt = torch.bfloat16
    g = torch.Generator(device='cpu').manual_seed(99)
    to = lambda t: t.to(gpu).to(tdt).contiguous(memory_format=torch.channels_last)
    xd, dyd = to(torch.randn(N, C, H, W, generator=g)), to(torch.randn(N, C, H, W, generator=g))
    gamma, beta = torch.ones(C, device=gpu), torch.zeros(C, device=gpu)
    rm, rv, nbt = torch.zeros(C, device=gpu), torch.ones(C, device=gpu), torch.zeros((), dtype=torch.int64, device=gpu)
    y, mean, invstd = ops.bn_train_fwd(xd, None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, False)

    def bwd():
        dg, db = torch.zeros(C, device=gpu), torch.zeros(C, device=gpu)
        dx, _ = ops.bn_bwd(dyd, xd, None, gamma, mean, invstd, dg, db, False, False, False, beta=beta)
        torch.cuda.synchronize()
        return dx, dg, db
    good = bwd()
    assert ops.bn_resident_timeouts() == 0 and all(bool(torch.isfinite(t.float()).all()) for t in good)
    prev = mi355.load().mi355_bn_set_resident(1)
    try:
        ops.bn_resident_set_spin_limit(1)
        dx, dg, db = bwd()
        n = ops.bn_resident_timeouts()
        assert n > 0, 'the shrunk spin bound did not provoke a give-up'
        assert bool(torch.isnan(dx.float()).any()), 'a block that gave up must poison its dx rows'
        with pytest.raises(mi355.Mi355Error, match='give-up'):
            ops.bn_resident_check('test')
        assert ops.bn_resident_timeouts() == 0                       # cleared by the check
        assert mi355.load().mi355_bn_set_resident(1) == 0            # ... which also switched the one-launch form off
    finally:
        ops.bn_resident_set_spin_limit(0)
        ops.bn_resident_reset()
        mi355.load().mi355_bn_set_resident(prev)
    again = bwd()
    assert ops.bn_resident_timeouts() == 0
    for a, b in zip(again, good):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 64, 32, 32), (3, 64, 18, 14)])
def test_stem_batchnorm_relu_maxpool_in_one_pass(gpu, dt, shape):
    """mi355_bn_relu_maxpool_fwd_partials == BatchNorm (conv-fused statistics) + ReLU, then MaxPool2d(3, 2, 1): pooled values,
    arg-max bytes, saved statistics and running statistics bit for bit."""
    ops = _ops()
    N, C, H, W = shape
    k, Ci = 3, 16
    xin = _nhwc(_round(randn(91, N, Ci, H, W), dt), dt, gpu)
    wm = (0.2 * randn(92, C, k, k, Ci)).to(gpu)
    wf, _ = ops.pack_weights(wm, C, k * k, Ci, Ci, DT[dt])
    desc = ops.make_desc(N, H, W, Ci, C, k, k, 1, 1, DT[dt])
    y, part = ops.conv_fwd_stats(desc, xin, wf, None)
    gamma, beta = (1 + 0.1 * randn(93, C)).to(gpu), (0.1 * randn(94, C)).to(gpu)
    res = []
    for fused in (False, True):
        rm, rv = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        if fused:
            p, arg, mean, invstd = ops.bn_relu_maxpool_fwd(y, gamma, beta, rm, rv, nbt, 1e-5, 0.1, 1, part)
        else:
            z, mean, invstd = ops.bn_train_fwd(y, None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, True, partial=part)
            p, arg = ops.maxpool_fwd(z)
        res.append((p, arg, mean, invstd, rm, rv, nbt))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_maxpool_and_colsum(gpu, dt):
    ops = _ops()
    x = _round(randn(31, 2, 64, 18, 14), dt)
    x[:, :, :6, :6] = 0.0            # ties: whole windows of equal values (post-ReLU zeros)
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    dy = _round(randn(32, *y_ref.shape), dt)
    y_ref.backward(dy)
    xd = _nhwc(x, dt, gpu)
    y, arg = ops.maxpool_fwd(xd)
    assert torch.equal(_back(y), y_ref.detach())
    dx = ops.maxpool_bwd(_nhwc(dy, dt, gpu), arg, x.shape)
    assert float((_back(dx) - xr.grad).abs().max()) <= _tol(dt, xr.grad)
    out = torch.zeros(64, device=gpu)
    ops.colsum(xd, out, False)
    ref = x.sum(dim=(0, 2, 3))
    assert float((out.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max() + 1)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_pointwise21(gpu, dt):
    ops = _ops()
    N, C, H, W, K = 2, 256, 16, 16, 21
    x = _round(randn(41, N, C, H, W), dt)
    w = randn(42, K, C, scale=0.06)
    b = randn(43, K, scale=0.1)
    xd = _nhwc(x, dt, gpu)
    y = ops.pw_c2k(xd, w.to(gpu), b.to(gpu), K)
    ref = F.conv2d(x, w.view(K, C, 1, 1), b)
    assert float((y.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
    # transposed-weight form = input gradient of the 21->C conv
    w2 = randn(44, C, K, scale=0.2)
    y2 = ops.pw_c2k(xd, w2.to(gpu), None, K, w_transposed=True)
    ref2 = F.conv2d(x, w2.t().contiguous().view(K, C, 1, 1))
    assert float((y2.cpu() - ref2).abs().max()) <= 1e-3 * float(ref2.abs().max())
    hm = randn(45, N, K, H, W)
    b2 = randn(46, C, scale=0.1)
    res = _round(randn(47, N, C, H, W), dt)
    out = ops.pw_k2c(hm.to(gpu), w2.to(gpu), b2.to(gpu), C, DT[dt], residual=_nhwc(res, dt, gpu))
    ref3 = F.conv2d(hm, w2.view(C, K, 1, 1), b2) + res
    assert float((_back(out) - ref3).abs().max()) <= _tol(dt, ref3)
    sc = torch.tensor(0.5, device=gpu)
    out2 = ops.pw_k2c(hm.to(gpu), w.to(gpu), None, C, DT[dt], scale_dev=sc, w_transposed=True)
    ref4 = 0.5 * F.conv2d(hm, w.t().contiguous().view(C, K, 1, 1))
    assert float((_back(out2) - ref4).abs().max()) <= _tol(dt, ref4)
    dw = torch.zeros(K, C, device=gpu)
    ops.pw_wgrad(xd, hm.to(gpu), dw, True, False)
    ref5 = torch.einsum('nkhw,nchw->kc', hm, x)
    assert float((dw.cpu() - ref5).abs().max()) <= 1e-3 * float(ref5.abs().max())
    dw2 = torch.zeros(C, K, device=gpu)
    ops.pw_wgrad(xd, hm.to(gpu), dw2, False, False)
    assert float((dw2.cpu() - ref5.t()).abs().max()) <= 1e-3 * float(ref5.abs().max())
    rs = torch.zeros(K, device=gpu)
    ops.hm_rowsum(hm.to(gpu), rs, False)
    assert torch.allclose(rs.cpu(), hm.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('hw', [(64, 64), (16, 16), (10, 12)])
def test_conv1x1_heatmap_mfma(gpu, dt, hw):
    """MFMA form of the C -> 21 heat-map conv (NCHW fp32 epilogue), incl. a map whose size is not a tile multiple."""
    ops = _ops()
    N, C, K = 3, 256, 21
    H, W = hw
    x = _round(randn(48, N, C, H, W), dt)
    w = _round(randn(49, K, C, scale=0.06), dt)
    b = randn(50, K, scale=0.1)
    y = ops.conv1x1_heatmap(_nhwc(x, dt, gpu), w.to(gpu).to(DT[dt]).contiguous(), b.to(gpu), K)
    ref = F.conv2d(x, w.view(K, C, 1, 1), b)
    assert y.dtype == torch.float32 and y.is_contiguous()
    assert float((y.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())


def test_argmax_bit_exact_and_accuracy(gpu):
    """Golden G4 (captured from the reference's numpy get_max_preds): integer indices bit-exact."""
    ops = _ops()
    from oracle import losses as ol
    g = golden('g4_argmax_accuracy')
    hm = peaky_heatmaps(401, 3, 21, 64, 64).numpy()
    hm[1, 0] = 0.5
    hm[1, 1, 10, 7] = hm[1, 1, 40, 3] = 9.0
    hm[2, 2, 63, 63] = 11.0
    idx, xy, mv = ops.argmax2d(torch.from_numpy(hm).to(gpu))
    assert np.array_equal(xy.cpu().numpy(), g['preds'])
    assert np.array_equal(mv.cpu().numpy(), g['maxvals'])
    flat = hm.reshape(3, 21, -1)
    assert np.array_equal(idx.cpu().numpy(), np.argmax(flat, 2).astype(np.int32))
    # NaN counts as the maximum (np.argmax), coordinates then masked to 0 (np.greater(nan, 0) is False)
    hn = hm.copy(); hn[0, 3, 5, 9] = np.nan
    idx2, xy2, _ = ops.argmax2d(torch.from_numpy(hn).to(gpu))
    p_ref, _ = ol.get_max_preds(hn)
    assert int(idx2[0, 3]) == 5 * 64 + 9 and np.array_equal(xy2.cpu().numpy(), p_ref)
    # ragged / non-square maps
    h2 = randn(402, 2, 5, 7, 11).numpy()
    _, xy3, mv3 = ops.argmax2d(torch.from_numpy(h2).to(gpu))
    p3, m3 = ol.get_max_preds(h2)
    assert np.array_equal(xy3.cpu().numpy(), p3) and np.array_equal(mv3.cpu().numpy(), m3)


def test_softargmax(gpu):
    ops = _ops()
    hm = randn(501, 2, 21, 64, 64, scale=0.05)
    hm[0, 0, 20, 33] += 1.0
    uv = ops.softargmax(hm.to(gpu))
    np.testing.assert_allclose(uv.cpu().numpy(), golden('g5_softargmax')['uv'], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('eps', [0.0, 1e-7])
def test_kl_heatmap_vs_oracle(gpu, eps):
    ops = _ops()
    from oracle import losses as ol
    B, K = 2, 21
    pred = randn(202, B, K, 64, 64)
    label = rand(205, B, K, 64, 64) * (rand(206, B, K, 64, 64) > 0.9)
    w = weights_bk(207, B, K)
    p = pred.clone().requires_grad_(True)
    ref = ol.JointsKLLoss(epsilon=eps)(p, label, w)
    ref.backward()
    rows, g = ops.kl_heatmap(pred.to(gpu), label.to(gpu), w.to(gpu), eps, True)
    loss = ops.reduce_sum(rows.view(-1), 1.0 / (B * K))
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref))
    assert float((g.cpu() - p.grad).abs().max()) <= 1e-3 * float(p.grad.abs().max())
    g2 = ops.scale_by_dev(g, torch.tensor(4.0, device=gpu))
    assert torch.allclose(g2.cpu(), 4 * g.cpu())


def _patch(tmp, sigma=2):
    from oracle.losses import _gauss_patch
    return torch.from_numpy(_gauss_patch(tmp, sigma).astype(np.float32).reshape(-1))


def test_pseudo_labels_bit_exact(gpu):
    """Golden G3 (reference PseudoLabelGenerator / 01 / 03 outputs): gt and gf bit-exact."""
    ops = _ops()
    g = golden('g3_pseudo_labels')
    y = peaky_heatmaps(301, 2, 21, 64, 64).to(gpu)
    _, xy, _ = ops.argmax2d(y)
    gt, gf = ops.pseudo_label(xy, _patch(6).to(gpu), 6, 1, 64, 0)
    assert np.array_equal(gt.cpu().numpy(), g['gt'])
    np.testing.assert_allclose(gf.cpu().numpy(), g['gf'], rtol=0, atol=2e-7)   # BLAS summation order in the reference
    gt1, gf1 = ops.pseudo_label(xy, _patch(3.0).to(gpu), 3, 4, 16, 1)
    assert np.array_equal(gt1.cpu().numpy(), g['gt01']) and np.array_equal(gf1.cpu().numpy(), g['gf01'])
    gt3, gf3 = ops.pseudo_label(xy, _patch(4).to(gpu), 4, 2, 32, 1)
    assert np.array_equal(gt3.cpu().numpy(), g['gt03']) and np.array_equal(gf3.cpu().numpy(), g['gf03'])


def test_ground_false_builders_vs_oracle(gpu):
    ops = _ops()
    from oracle import losses as ol
    import torch.nn as nn
    B, K = 2, 21
    y = peaky_heatmaps(201, B, K, 64, 64)
    y_adv2, y_adv3 = randn(203, B, K, 32, 32), randn(204, B, K, 16, 16)
    up = lambda t, s: nn.Upsample(size=s, mode='bilinear')(t)
    t5_ref, t0_ref = 0.5 * up(y_adv3, 64) + up(y_adv2, 64), up(y_adv3, 32)
    t5 = ops.bilinear_up(y_adv3.to(gpu), 64, 0.5)
    t5 = ops.bilinear_up(y_adv2.to(gpu), 64, 1.0, out=t5)
    t0 = ops.bilinear_up(y_adv3.to(gpu), 32)
    assert torch.allclose(t5.cpu(), t5_ref, rtol=1e-5, atol=1e-5) and torch.allclose(t0.cpu(), t0_ref, rtol=1e-5, atol=1e-5)
    g2 = golden('g2_losses')
    np.testing.assert_allclose(t5.cpu().numpy()[:, :2], g2['target5'], rtol=1e-5, atol=1e-5)
    _, xy, _ = ops.argmax2d(y.to(gpu))
    rd6 = ol.RegressionDisparityx6(ol.PseudoLabelGenerator(K, 64, 64), ol.JointsKLLoss(epsilon=1e-7))
    rd5 = ol.RegressionDisparityx5(ol.PseudoLabelGenerator03(K), ol.JointsKLLoss(epsilon=1e-7))
    w = weights_bk(207, B, K)
    for extra_ref, extra in ((None, None), (t5_ref, t5)):
        rd6(y, randn(1, B, K, 64, 64), extra_ref, w, mode='max')
        _, gf = ops.pseudo_label(xy, _patch(6).to(gpu), 6, 1, 64, 2, extra=extra, normalise=True)
        np.testing.assert_allclose(gf.cpu().numpy(), rd6.ground_false.numpy(), rtol=1e-5, atol=1e-6)
    for extra_ref, extra in ((None, None), (t0_ref, t0)):
        rd5(y, randn(1, B, K, 32, 32), extra_ref, w, mode='max')
        _, gf = ops.pseudo_label(xy, _patch(4).to(gpu), 4, 2, 32, 1, extra=extra, normalise=True)
        np.testing.assert_allclose(gf.cpu().numpy(), rd5.ground_false.numpy(), rtol=1e-5, atol=1e-6)


def test_sgd_nesterov_vs_torch(gpu):
    ops = _ops()
    n = 10007
    p0, g1, g2 = randn(61, n), randn(62, n), randn(63, n)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    for g, lr in ((g1, 1e-3), (g2, 7e-4)):
        opt.param_groups[0]['lr'] = lr
        p.grad = g.clone()
        opt.step()
    pad = (n + 3) // 4 * 4
    pd = torch.zeros(pad, device=gpu); pd[:n] = p0.to(gpu)
    buf = torch.zeros(pad, device=gpu); gd = torch.zeros(pad, device=gpu)
    low = torch.zeros(pad, dtype=torch.bfloat16, device=gpu)
    lr_dev = torch.zeros((), device=gpu)
    for g, lr in ((g1, 1e-3), (g2, 7e-4)):
        gd[:n] = g.to(gpu); lr_dev.fill_(lr)
        ops.sgd_nesterov(pd[:n], gd[:n], buf[:n], lr_dev, 0.9, 1e-4, True, low[:n])
    assert torch.allclose(pd[:n].cpu(), p.detach(), rtol=1e-6, atol=1e-7)
    assert torch.equal(low[:n].cpu(), pd[:n].cpu().to(torch.bfloat16))


def test_errors_are_loud(gpu):
    import mi355
    ops = _ops()
    with pytest.raises(mi355.Mi355Error):
        ops.to_nhwc(torch.zeros(1, 3, 4, 4))          # CPU tensor: no fallback
    desc = ops.make_desc(1, 8, 8, 24, 64, 3, 3, 1, 1, torch.bfloat16)   # Ci/8 = 3 is not a power of two
    x = torch.zeros(1, 8, 8, 24, dtype=torch.bfloat16, device=gpu).permute(0, 3, 1, 2)
    with pytest.raises(mi355.Mi355Error):
        ops.conv_fwd(desc, x, torch.zeros(64 * 9 * 24, dtype=torch.bfloat16, device=gpu))


def test_device_label_generation_matches_generate_target(gpu):
    """utils.labels.generate_target_device vs the oracle's restatement of uda/dataset/util.py:9-68, bit-exact, incl.
    centres on the border, outside the map, and invisible joints."""
    from utils.labels import generate_target_device
    from oracle.losses import generate_target
    rng = np.random.default_rng(5)
    B, K = 3, 21
    kp = rng.uniform(-12, 268, size=(B, K, 2))
    kp[0, 0] = (0.0, 0.0); kp[0, 1] = (255.9, 255.9); kp[0, 2] = (3.9, 250.0); kp[0, 3] = (258.0, 10.0); kp[0, 4] = (-1.9, 30.0)
    vis = (rng.random((B, K, 1)) < 0.8).astype(np.float32)
    ref_t = np.zeros((B, K, 64, 64), np.float32); ref_w = np.zeros((B, K, 1), np.float32)
    for b in range(B):
        ref_t[b], ref_w[b] = generate_target(kp[b], vis[b], (64, 64), 2, (256, 256))
    t, w = generate_target_device(torch.from_numpy(kp).to(gpu), torch.from_numpy(vis).to(gpu))
    assert np.array_equal(w.cpu().numpy(), ref_w)
    assert np.array_equal(t.cpu().numpy(), ref_t)
    # and against the arrays the reference's own generate_target produced (golden G9), at three resolutions
    from seeded import g9_inputs
    g = golden('g9_generate_target')
    kp, vis = g9_inputs()
    for tag, (hm, img) in dict(a=(64, 256), b=(32, 128), c=(128, 512)).items():
        t, w = generate_target_device(torch.from_numpy(kp * (img / 256.0)).to(gpu), torch.from_numpy(vis).to(gpu), hm, 2, img)
        assert np.array_equal(w.cpu().numpy(), g['weight_' + tag]), tag
        assert np.array_equal(t.cpu().numpy(), g['target_' + tag]), tag


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['none', 'mask_x', 'mask_y'])
@pytest.mark.parametrize('case', [('conv', 2, 64, 16, 16, 128, 3, 1, 1, False), ('conv', 2, 64, 16, 16, 128, 3, 2, 1, True),
                                  ('conv', 3, 128, 8, 8, 256, 1, 1, 0, False), ('conv', 2, 64, 16, 16, 256, 1, 2, 0, False),
                                  ('deconv', 2, 256, 16, 16, 64, 4, 2, 1, False)])
def test_gemm_epilogue_bn_backward_reduction(gpu, dt, mode, case):
    """dgrad (or the deconv input gradient) that also reduces its result for the BatchNorm it feeds == separate passes."""
    ops = _ops()
    kind, N, Ci, H, W, Co, k, s, p, acc = case
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, DT[dt])
    w = _round(randn(2, Co, Ci, k, k, scale=1.0 / np.sqrt(Ci * k * k)), dt)
    wf, wt = ops.pack_weights(w.permute(0, 2, 3, 1).contiguous().to(gpu), Co, k * k, Ci, Ci, DT[dt])
    if kind == 'conv':       # the GEMM output has the conv INPUT's shape
        g_in = _nhwc(_round(randn(1, N, Co, desc.Ho, desc.Wo), dt), dt, gpu)
        C, Hh, Ww = Ci, H, W
    else:
        g_in = _nhwc(_round(randn(1, N, Ci, H, W), dt), dt, gpu)
        C, Hh, Ww = Co, desc.Ho, desc.Wo
    xb = _nhwc(_round(randn(5, N, C, Hh, Ww) * 1.3 + 0.2, dt), dt, gpu)        # the BatchNorm's forward input
    gamma, beta = (1 + 0.1 * randn(23, C)).to(gpu), (0.1 * randn(24, C)).to(gpu)
    rm, rv, nbt = torch.zeros(C, device=gpu), torch.ones(C, device=gpu), torch.zeros((), dtype=torch.int64, device=gpu)
    relu = mode != 'none'
    yb, mean, invstd = ops.bn_train_fwd(xb, None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, relu)
    y_mask = yb if mode == 'mask_y' else None
    base = _nhwc(_round(randn(6, N, C, Hh, Ww), dt), dt, gpu) if acc else None
    bn = (xb, y_mask, gamma, beta, mean, invstd, relu)
    if kind == 'conv':
        dy0 = ops.conv_dgrad(desc, g_in, wt, out=base.clone() if acc else None, accumulate=acc)
        dy1, part = ops.conv_dgrad_bnbwd(desc, g_in, wt, bn, out=base.clone() if acc else None, accumulate=acc)
    else:
        dy0 = ops.conv_fwd(desc, g_in, wf)
        dy1, part = ops.conv_fwd_bnbwd(desc, g_in, wf, bn)
    assert part is not None and torch.equal(dy0, dy1)
    res = []
    for pp in (None, part):
        dg, db = torch.zeros(C, device=gpu), torch.zeros(C, device=gpu)
        dx, _ = ops.bn_bwd(dy1, xb, y_mask, gamma, mean, invstd, dg, db, False, relu, False, beta=beta, partial=pp)
        res.append((dx, dg, db))
    (dxa, dga, dba), (dxb, dgb, dbb) = res
    sc = float(dga.abs().max()) + float(dba.abs().max()) + 1.0
    assert float((dga - dgb).abs().max()) <= 2e-5 * sc and float((dba - dbb).abs().max()) <= 2e-5 * sc
    assert float((dxa.float() - dxb.float()).abs().max()) <= (2e-2 if dt == 'bf16' else 1e-4) * (float(dxa.float().abs().max()) + 1e-6)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_pack_weights_batched(gpu, dt):
    """One launch packing several conv weights == the per-weight pack (incl. the padded 3-channel stem)."""
    ops = _ops()
    shapes = [(64, 49, 3, 8 if dt == 'bf16' else 4), (128, 9, 64, 64), (40, 1, 72, 72), (256, 16, 64, 64)]    # O, T, I, Ipad
    ws = [randn(30 + i, O, T, I).to(gpu) for i, (O, T, I, Ip) in enumerate(shapes)]
    ref = [ops.pack_weights(w, O, T, I, Ip, DT[dt]) for w, (O, T, I, Ip) in zip(ws, shapes)]
    outs = [(torch.full_like(a, 3.0), torch.full_like(b, 3.0)) for a, b in ref]
    rec = np.zeros(len(shapes), dtype=[('w', '<u8'), ('wf', '<u8'), ('wt', '<u8'), ('O', '<i4'), ('T', '<i4'), ('I', '<i4'),
                                       ('Ipad', '<i4'), ('blk0', '<i4'), ('pad', '<i4')])
    blk = 0
    for i, ((O, T, I, Ip), w, (wf, wt)) in enumerate(zip(shapes, ws, outs)):
        rec[i] = (w.data_ptr(), wf.data_ptr(), wt.data_ptr(), O, T, I, Ip, blk, 0)
        blk += ((Ip + 31) // 32) * ((O + 31) // 32) * T
    tab = torch.from_numpy(rec.view(np.uint8).copy()).to(gpu)
    ops.pack_weights_batched(tab, len(shapes), blk, DT[dt])
    for (a, b), (wf, wt) in zip(ref, outs):
        assert torch.equal(a, wf) and torch.equal(b, wt)


def test_fused_statistics_fall_back_when_phases_are_uneven(gpu):
    """A transposed conv onto an odd-sized map has output phases of different tile counts: the launch declines to fuse the
    statistics (nslices = 0 -> None) and still produces the plain result; BatchNorm then runs its own statistics pass."""
    ops = _ops()
    N, Ci, H, W, Co, k, s, p = 1, 64, 17, 17, 64, 3, 2, 1          # conv-form: input 17x17 -> output 9x9; deconv output = 17x17
    desc = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, torch.float32)
    w = randn(2, Co, Ci, k, k, scale=0.05)
    wf, wt = ops.pack_weights(w.permute(0, 2, 3, 1).contiguous().to(gpu), Co, k * k, Ci, Ci, torch.float32)
    x = _nhwc(randn(1, N, Co, desc.Ho, desc.Wo), 'f32', gpu)
    y, part = ops.conv_dgrad_stats(desc, x, wt)
    assert torch.equal(y, ops.conv_dgrad(desc, x, wt))
    if part is not None:                                           # (fusing is allowed whenever the tile counts happen to match)
        assert part[1] >= 1
    C = y.shape[1]
    outs = [ops.bn_train_fwd(y, None, torch.ones(C, device=gpu), torch.zeros(C, device=gpu), torch.zeros(C, device=gpu),
                             torch.ones(C, device=gpu), torch.zeros((), dtype=torch.int64, device=gpu), 1e-5, 0.1, True, partial=pp)[0]
            for pp in (None, part)]
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-5


def test_guarded_max_normalisation_keeps_empty_maps_at_zero(gpu):
    """normalise=2 (benchmark extension): a ground-false map whose maximum is 0 stays 0; normalise=1 keeps the reference's
    0/0 = NaN; maps with a positive maximum are identical in both modes."""
    from uda.model.regda_7 import RegressionDisparityx6, PseudoLabelGenerator
    from uda.model.loss import JointsKLLoss
    B, K, S = 2, 21, 64
    y = torch.full((B, K, S, S), -1.0, device=gpu)
    for k in range(K):                       # image 0: every key-point peaks at the same pixel; image 1: spread out
        y[0, k, 30, 30] = 5.0
        y[1, k, 5 + 2 * k, 7 + 2 * k] = 5.0
    extra = torch.zeros(B, K, S, S, device=gpu)
    extra[0] = -3.0                                            # a very negative y_adv2 term empties the maps of image 0
    y_adv = torch.randn(B, K, S, S, device=gpu)
    w = torch.ones(B, K, 1, device=gpu)
    outs = {}
    for guard in (False, True):
        rd = RegressionDisparityx6(PseudoLabelGenerator(K, S, S), JointsKLLoss(epsilon=1e-7))
        rd.guard_empty_maps = guard
        loss = rd(y, y_adv, extra, w, mode='max')
        outs[guard] = (rd.ground_false.clone(), float(loss))
    gf_ref, l_ref = outs[False]
    gf_g, l_g = outs[True]
    assert torch.isnan(gf_ref[0]).any() and l_ref != l_ref                 # the reference rule: NaN
    assert torch.isfinite(gf_g).all() and l_g == l_g
    assert float(gf_g[0].abs().max()) == 0.0
    assert torch.equal(gf_ref[1], gf_g[1])                                  # non-empty maps: same result


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 21, 64, 256, 16, 16), (3, 21, 60, 256, 10, 10), (2, 21, 64, 128, 8, 8)])
def test_pointwise_k2c_epilogue_bn_statistics(gpu, dt, shape):
    """21 -> C point-wise conv (+bias, +residual) with the BatchNorm statistics of its output from the epilogue == stand-alone pass."""
    ops = _ops()
    N, K, _, C, H, W = shape
    hm = randn(41, N, K, H, W).to(gpu)
    w = (0.2 * randn(42, C, K)).to(gpu)
    b = (0.1 * randn(43, C)).to(gpu)
    res = _nhwc(_round(randn(44, N, C, H, W), dt), dt, gpu)
    y0 = ops.pw_k2c(hm, w, b, C, DT[dt], residual=res)
    y1, part = ops.pw_k2c_stats(hm, w, b, C, DT[dt], residual=res)
    assert torch.equal(y0, y1) and part[1] == N * ((H * W + 63) // 64)
    gamma, beta = (1 + 0.1 * randn(23, C)).to(gpu), (0.1 * randn(24, C)).to(gpu)
    outs = []
    for pp in (None, part):
        rm, rv = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        outs.append(ops.bn_train_fwd(y1, None, gamma, beta, rm, rv, nbt, 1e-5, 0.1, True, partial=pp) + (rm, rv))
    (ya, ma, ia, rma, rva), (yb, mb, ib, rmb, rvb) = outs
    assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-6) and torch.allclose(ia, ib, rtol=2e-5, atol=1e-6)
    assert torch.allclose(rma, rmb, rtol=1e-5, atol=1e-6) and torch.allclose(rva, rvb, rtol=2e-5, atol=1e-6)
    assert float((ya.float() - yb.float()).abs().max()) <= (2e-2 if dt == 'bf16' else 1e-4)


def test_pointwise_k2c_walks_several_tiles_per_block_at_full_size(gpu):
    """The benchmark's 64 x 21 x 64 x 64 heat-maps: 4096 pixel tiles on at most 1024 persistent blocks, so every block walks
    several tiles with the weights it loaded once.  Output against torch on the same operands, statistics partials against a
    per-slice torch computation on the stored (rounded) values (slice = 64 consecutive pixels of one image)."""
    ops = _ops()
    N, K, C, H, W = 64, 21, 256, 64, 64
    g = torch.Generator(device='cpu').manual_seed(77)
    hm = torch.randn(N, K, H, W, generator=g).to(gpu)
    w = (0.2 * torch.randn(C, K, generator=g)).to(gpu)
    b = (0.1 * torch.randn(C, generator=g)).to(gpu)
    res = ops.nhwc_empty(N, C, H, W, torch.bfloat16, gpu).normal_()
    y0 = ops.pw_k2c(hm, w, b, C, torch.bfloat16, residual=res)
    y1, part = ops.pw_k2c_stats(hm, w, b, C, torch.bfloat16, residual=res)
    assert torch.equal(y0, y1) and part[1] == N * (H * W // 64)
    ref = torch.einsum('nkhw,ck->nchw', hm, w) + b.view(1, -1, 1, 1) + res.float()
    assert float((y0.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    rows = y1.permute(0, 2, 3, 1).reshape(N * H * W // 64, 64, C).float()            # [slice][pixel][channel]
    p = part[0][:part[1] * C * 3].view(part[1], C, 3)
    assert torch.equal(p[:, :, 0], torch.full_like(p[:, :, 0], 64.0))
    mean = rows.mean(dim=1)
    m2 = ((rows - mean[:, None, :]) ** 2).sum(dim=1)
    assert float((p[:, :, 1] - mean).abs().max()) <= 1e-5 * float(mean.abs().max() + 1)
    assert float((p[:, :, 2] - m2).abs().max()) <= 1e-4 * float(m2.abs().max() + 1)


def test_grouped_weight_gradients_of_a_resnet_stage_at_full_size(gpu):
    """One wgrad_group launch of layer3's 1x1 convs at the benchmark's size (B=64, 16x16 maps, bf16), every problem against
    torch's fp32 weight gradient on the same bf16-rounded operands (round-2 verdict, item 8)."""
    ops = _ops()
    tdt = torch.bfloat16
    shapes = [(64, 16, 16, 1024, 256, 1, 1, 0), (64, 16, 16, 256, 1024, 1, 1, 0), (64, 32, 32, 512, 1024, 1, 2, 0), (64, 32, 32, 512, 256, 1, 1, 0)] * 2
    items, refs = [], []
    for i, (N, H, W, Ci, Co, k, s, p) in enumerate(shapes):
        d = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, tdt)
        x = ops.nhwc_empty(N, Ci, H, W, tdt, gpu).normal_()
        dy = ops.nhwc_empty(N, Co, d.Ho, d.Wo, tdt, gpu).normal_()
        xs = x[:, :, ::s, ::s].permute(0, 2, 3, 1).reshape(-1, Ci).float()           # 1x1: dW[o][c] = sum_m dy[m][o] x[m*s][c]
        refs.append((dy.permute(0, 2, 3, 1).reshape(-1, Co).float().t() @ xs).reshape(-1))
        items.append((d, x, dy, torch.full((Co * Ci,), float('nan'), device=gpu), False))
    ops.conv_wgrad_grouped(items)
    torch.cuda.synchronize()
    for i, ((d, x, dy, dw, acc), ref) in enumerate(zip(items, refs)):
        assert torch.isfinite(dw).all(), i
        assert float((dw - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), (i, shapes[i])


def test_grouped_weight_gradients_on_256_wide_tiles(gpu):
    """wgrad_group256_kernel (256 x 256 output tiles, 8 waves; the 1x1 convs with >= 256 channels on both sides): layer4's shapes at
    the benchmark's size -- 512 <-> 2048 @8x8, the strided 1024 -> 2048 down-sample -- and a head's 256 -> 256 @16x16, some items
    accumulating onto a prefilled gradient, mixed with an item the kernel does not take (128 -> 512: 128-wide columns) -- every
    problem against torch's fp32 weight gradient on the same bf16-rounded operands."""
    ops = _ops()
    tdt = torch.bfloat16
    shapes = [(64, 8, 8, 2048, 512, 1, 1, 0, False), (64, 8, 8, 512, 2048, 1, 1, 0, True), (64, 16, 16, 1024, 2048, 1, 2, 0, False),
              (64, 32, 32, 128, 512, 1, 1, 0, False), (64, 16, 16, 256, 256, 1, 1, 0, True), (3, 9, 7, 256, 512, 1, 1, 0, False)]
    items, refs = [], []
    for i, (N, H, W, Ci, Co, k, s, p, acc) in enumerate(shapes):
        d = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, tdt)
        x = ops.nhwc_empty(N, Ci, H, W, tdt, gpu).normal_()
        dy = ops.nhwc_empty(N, Co, d.Ho, d.Wo, tdt, gpu).normal_()
        xs = x[:, :, ::s, ::s].permute(0, 2, 3, 1).reshape(-1, Ci).float()
        ref = (dy.permute(0, 2, 3, 1).reshape(-1, Co).float().t() @ xs).reshape(-1)
        dw = torch.full((Co * Ci,), 0.5 if acc else float('nan'), device=gpu)
        refs.append(ref + 0.5 if acc else ref)
        items.append((d, x, dy, dw, acc))
    ops.conv_wgrad_grouped(items)
    torch.cuda.synchronize()
    for i, ((d, x, dy, dw, acc), ref) in enumerate(zip(items, refs)):
        assert torch.isfinite(dw).all(), i
        assert float((dw - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), (i, shapes[i])


def test_grouped_3x3_weight_gradients_of_resnet_stages_at_full_size(gpu):
    """The 3x3 / stride-1 weight gradients of a ResNet stage in one wgrad_kw_group launch (B=64, bf16): layer3's six 256 -> 256
    @16x16, layer1's 64 -> 64 @64x64 (one output-channel tile: its own group), layer4's 512 -> 512 @8x8, layer2's 128 -> 128 @32x32,
    with accumulating items and an item that writes the gradient of an earlier one -- against mi355_conv_wgrad problem by problem,
    and two runs bit for bit."""
    ops = _ops()
    tdt = torch.bfloat16
    shapes = [(64, 16, 16, 256, 256)] * 6 + [(64, 64, 64, 64, 64)] * 3 + [(64, 8, 8, 512, 512)] * 2 + [(64, 32, 32, 128, 128)] * 3
    items, refs, bases = [], [], []
    for i, (N, H, W, Ci, Co) in enumerate(shapes):
        d = ops.make_desc(N, H, W, Ci, Co, 3, 3, 1, 1, tdt)
        x = ops.nhwc_empty(N, Ci, H, W, tdt, gpu).normal_()
        dy = ops.nhwc_empty(N, Co, H, W, tdt, gpu).normal_()
        acc = (i % 4 == 1)
        base = torch.randn(Co * 9 * Ci, device=gpu) if acc else None
        ref = base.clone() if acc else torch.empty(Co * 9 * Ci, device=gpu)
        ops.conv_wgrad(d, x, dy, ref, acc)
        items.append((d, x, dy, torch.empty(Co * 9 * Ci, device=gpu), acc)); refs.append(ref); bases.append(base)
    # the first layer3 conv once more, accumulating onto its own gradient (overwrite, then accumulate: the order must hold)
    d0, x0, dy0, dw0, _ = items[0]
    x0b, dy0b = torch.randn_like(x0), torch.randn_like(dy0)
    ops.conv_wgrad(d0, x0b, dy0b, refs[0], True)
    items.append((d0, x0b, dy0b, dw0, True))
    outs = []
    for _ in range(2):
        for (d, x, dy, dw, acc), base in zip(items[:-1], bases):
            dw.copy_(base) if acc else dw.fill_(float('nan'))
        ops.conv_wgrad_grouped(items)
        torch.cuda.synchronize()
        outs.append([it[3].clone() for it in items[:-1]])
    for i, (dw, ref) in enumerate(zip(outs[0], refs)):
        assert torch.isfinite(dw).all(), i
        assert float((dw - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), (i, shapes[i])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_grouped_weight_gradients_keep_the_order_of_items_that_share_a_gradient(gpu):
    """One conv used twice in a backward hands the group two items with the SAME dw (overwrite, then accumulate): they must not
    share a launch (the overwrite and the read-modify-write would race) and must keep their order (advisor, round 2)."""
    ops = _ops()
    tdt = torch.bfloat16
    N, H, W, Ci, Co = 4, 16, 16, 64, 256
    d = ops.make_desc(N, H, W, Ci, Co, 1, 1, 1, 0, tdt)
    xs = [ops.nhwc_empty(N, Ci, H, W, tdt, gpu).normal_() for _ in range(2)]
    dys = [ops.nhwc_empty(N, Co, H, W, tdt, gpu).normal_() for _ in range(2)]
    other = ops.make_desc(2, 8, 8, 128, 512, 1, 1, 1, 0, tdt)
    xo, dyo = ops.nhwc_empty(2, 128, 8, 8, tdt, gpu).normal_(), ops.nhwc_empty(2, 512, 8, 8, tdt, gpu).normal_()
    dw = torch.full((Co * Ci,), float('nan'), device=gpu)
    dwo = torch.full((512 * 128,), float('nan'), device=gpu)
    ref = torch.empty(Co * Ci, device=gpu)
    ops.conv_wgrad(d, xs[0], dys[0], ref, False); ops.conv_wgrad(d, xs[1], dys[1], ref, True)
    for _ in range(3):
        dw.fill_(float('nan'))
        ops.conv_wgrad_grouped([(d, xs[0], dys[0], dw, False), (other, xo, dyo, dwo, False), (d, xs[1], dys[1], dw, True)])
        torch.cuda.synchronize()
        assert torch.isfinite(dw).all()
        assert float((dw - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_grouped_weight_gradients_match_single_launches(gpu, dt):
    """mi355_conv_wgrad_grouped: many small problems in one launch (split counts chosen for the group, grouped slab
    reduction) against mi355_conv_wgrad problem by problem -- same sums up to the fp32 summation order -- including
    accumulate=1, a strided 1x1, a 3x3 that goes to the specialised kernel, and more problems than one group holds."""
    ops = _ops()
    tdt = torch.float32 if dt == 'f32' else torch.bfloat16
    shapes = [(4, 16, 16, 64, 256, 1, 1, 0), (4, 16, 16, 256, 64, 1, 1, 0), (2, 8, 8, 512, 128, 1, 1, 0), (4, 16, 16, 64, 128, 1, 2, 0),
              (2, 16, 16, 64, 64, 3, 1, 1), (3, 8, 8, 128, 512, 1, 1, 0), (2, 16, 16, 32, 64, 3, 2, 1)] * 4          # 28 problems
    items, refs = [], []
    for i, (N, H, W, Ci, Co, k, s, p) in enumerate(shapes):
        d = ops.make_desc(N, H, W, Ci, Co, k, k, s, p, tdt)
        x = ops.nhwc_empty(N, Ci, H, W, tdt, gpu).copy_(randn(100 + i, N, Ci, H, W).to(gpu))
        dy = ops.nhwc_empty(N, Co, d.Ho, d.Wo, tdt, gpu).copy_(randn(200 + i, N, Co, d.Ho, d.Wo).to(gpu))
        acc = (i % 3 == 0)
        base = randn(300 + i, Co * k * k * Ci).to(gpu)
        dw = base.clone() if acc else torch.full((Co * k * k * Ci,), float('nan'), device=gpu)
        ref = base.clone() if acc else torch.empty(Co * k * k * Ci, device=gpu)
        ops.conv_wgrad(d, x, dy, ref, acc)
        items.append((d, x, dy, dw, acc)); refs.append(ref)
    ops.conv_wgrad_grouped(items)
    torch.cuda.synchronize()
    for i, ((d, x, dy, dw, acc), ref) in enumerate(zip(items, refs)):
        assert torch.isfinite(dw).all(), i
        scale = float(ref.abs().max())
        assert float((dw - ref).abs().max()) <= (2e-5 if dt == 'f32' else 1e-4) * scale, (i, shapes[i])
