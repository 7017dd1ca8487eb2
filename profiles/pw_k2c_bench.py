"""pw_k2c (21 -> 256 channels) in isolation: plain, + residual, + residual + statistics; md5 of the output for bit-identity checks
across builds (MI355_LIB=...).  python profiles/pw_k2c_bench.py [tag]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'domain-adaptative-hand-pose-estimation_amd'), os.path.join(ROOT, 'profiles')]
import mi355
from mi355 import ops
from pgemm_bench import timeit
mi355.load()
dev = torch.device("cuda:0"); torch.manual_seed(3)
tag = sys.argv[1] if len(sys.argv) > 1 else ''
for (N, H, C, K) in [(64, 64, 256, 21), (64, 32, 256, 21), (64, 64, 256, 42)]:
    if K > 32: continue
    y = torch.randn(N, K, H, H, device=dev)
    w = torch.randn(C, K, device=dev) * 0.1
    b = torch.randn(C, device=dev)
    res = ops.nhwc_empty(N, C, H, H, torch.bfloat16, dev).normal_()
    o1 = ops.pw_k2c(y, w, b, C, torch.bfloat16)
    o2 = ops.pw_k2c(y, w, b, C, torch.bfloat16, residual=res)
    o3, _ = ops.pw_k2c_stats(y, w, b, C, torch.bfloat16, residual=res)
    ref = torch.einsum('nkhw,ck->nchw', y, w) + b.view(1, -1, 1, 1)
    e1 = float((o1.float() - ref).abs().max())
    import hashlib
    h = hashlib.md5(o2.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:8]
    t1 = timeit(lambda: ops.pw_k2c(y, w, b, C, torch.bfloat16))
    t2 = timeit(lambda: ops.pw_k2c(y, w, b, C, torch.bfloat16, residual=res))
    t3 = timeit(lambda: ops.pw_k2c_stats(y, w, b, C, torch.bfloat16, residual=res))
    print('%s k2c N%d H%d C%d K%d: plain %.1f us, +res %.1f us, +res+stats %.1f us  err %.2e md5 %s' % (tag, N, H, C, K, t1*1e6, t2*1e6, t3*1e6, e1, h), flush=True)
