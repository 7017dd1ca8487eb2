"""fp8 against bf16 weight-gradient kernels per layer shape (stride-1 and stride-2 3x3, 4x4), graph-timed in isolation.
python profiles/fp8_wgrad_bench.py [B]"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/domain-adaptative-hand-pose-estimation_amd', R + '/profiles']
import mi355
from mi355 import ops
from pgemm_bench import timeit
mi355.load(); dev = torch.device('cuda:0'); dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (H, C) in [(64, 256), (32, 256), (32, 128), (16, 256), (8, 512), (128, 64) if B <= 32 else (64, 64)]:
    d16 = ops.make_desc(B, H, H, C, C, 3, 3, 1, 1, dt)
    d8 = ops.make_desc_fp8(B, H, H, C, C, 3, 3, 1, 1)
    x = ops.nhwc_empty(B, C, H, H, dt, dev).normal_(); dy = ops.nhwc_empty(B, C, H, H, dt, dev).normal_().mul_(1e-2)
    sx, sd = ops.fp8_state(dev), ops.fp8_state(dev)
    x8 = ops.fp8_quantize(x, sx, ops.E4M3, jit=True); dy8 = ops.fp8_quantize(dy, sd, ops.E5M2, jit=True)
    dw = torch.zeros(C, 3, 3, C, device=dev); dw8 = torch.zeros(C, 3, 3, C, device=dev)
    ops.conv_wgrad(d16, x, dy, dw, False); ops.conv_wgrad_fp8(d8, x8, sx, dy8, sd, dw8, False)
    torch.cuda.synchronize()
    rel = float((dw - dw8).norm() / dw.norm())
    t16 = timeit(lambda: ops.conv_wgrad(d16, x, dy, dw, False)); t8 = timeit(lambda: ops.conv_wgrad_fp8(d8, x8, sx, dy8, sd, dw8, False))
    fl = 2.0 * B * H * H * C * C * 9
    print('wgrad 3x3 %d->%d @%d B=%d: bf16 %.1f us (%.0f TF/s)  fp8 %.1f us (%.0f TF/s)   fp8 vs bf16 result: rel L2 %.3f' % (C, C, H, B, t16 * 1e6, fl / t16 / 1e12, t8 * 1e6, fl / t8 / 1e12, rel), flush=True)

for (H, C, k) in [(64, 256, 3), (32, 256, 3), (64, 256, 4), (32, 256, 4), (16, 512, 3)]:
    d16 = ops.make_desc(B, H, H, C, C, k, k, 2, 1, dt)
    d8 = ops.make_desc_fp8(B, H, H, C, C, k, k, 2, 1)
    x = ops.nhwc_empty(B, C, H, H, dt, dev).normal_(); dy = ops.nhwc_empty(B, C, H // 2, H // 2, dt, dev).normal_().mul_(1e-2)
    sx, sd = ops.fp8_state(dev), ops.fp8_state(dev)
    x8 = ops.fp8_quantize(x, sx, ops.E4M3, jit=True); dy8 = ops.fp8_quantize(dy, sd, ops.E5M2, jit=True)
    dw = torch.zeros(C, k, k, C, device=dev); dw8 = torch.zeros(C, k, k, C, device=dev)
    ops.conv_wgrad(d16, x, dy, dw, False); ops.conv_wgrad_fp8(d8, x8, sx, dy8, sd, dw8, False)
    torch.cuda.synchronize()
    rel = float((dw - dw8).norm() / dw.norm())
    t16 = timeit(lambda: ops.conv_wgrad(d16, x, dy, dw, False)); t8 = timeit(lambda: ops.conv_wgrad_fp8(d8, x8, sx, dy8, sd, dw8, False))
    fl = 2.0 * B * (H // 2) ** 2 * C * C * k * k
    print('wgrad %dx%d s2 %d->%d @%d B=%d: bf16 %.1f us (%.0f TF/s)  fp8 %.1f us (%.0f TF/s)   fp8 vs bf16 result: rel L2 %.3f' % (k, k, C, C, H, B, t16 * 1e6, fl / t16 / 1e12, t8 * 1e6, fl / t8 / 1e12, rel), flush=True)
