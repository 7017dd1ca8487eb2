"""1x1 / unit-stride conv layers of the ResNet-50 iteration in isolation: time per launch and a correctness check against a
torch matmul, for whichever kernel the library picks under the current environment
(MI355_PGEMM=0: gather kernel of igemm.hip; default: persistent pipelined GEMM of pgemm.hip; MI355_PG_TILE=n: force a tile).

    python profiles/pgemm_bench.py [tag]        -> one line per (layer, kind): us, TFLOP/s, GB/s of algorithmic bytes, max error
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'domain-adaptative-hand-pose-estimation_amd')]
import mi355  # noqa: E402
from mi355 import ops  # noqa: E402

dev = torch.device('cuda:0')
mi355.load()
dt = torch.bfloat16
# (N, H, Ci, Co, launches per iteration fwd, dgrad)  -- profiles/r02_layer_table.txt
LAYERS = [(64, 64, 64, 256, 8, 8), (64, 16, 256, 1024, 12, 12), (64, 32, 128, 512, 8, 8), (64, 16, 1024, 256, 10, 10),
          (64, 64, 256, 256, 3, 2), (64, 32, 512, 128, 6, 6), (64, 64, 256, 64, 4, 4), (64, 8, 512, 2048, 6, 6),
          (64, 8, 2048, 512, 4, 4), (64, 64, 256, 128, 2, 2), (64, 32, 256, 256, 3, 3), (64, 32, 512, 256, 2, 2),
          (64, 16, 1024, 512, 2, 2), (64, 64, 64, 64, 2, 2), (64, 16, 256, 256, 3, 2)]


def timeit(fn, n=20):
    """seconds per launch with the host out of the picture: n launches captured in a HIP graph, the graph replayed 5 times
    (eager launches from Python are host-bound below ~15 us per call and hide every difference between kernels)"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e-3


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ''
    tot = {'fwd': 0.0, 'fwd_stats': 0.0, 'dgrad': 0.0}
    g = torch.Generator(device='cpu').manual_seed(5)
    sel = [int(i) for i in os.environ['PG_LAYERS'].split(',')] if os.environ.get('PG_LAYERS') else range(len(LAYERS))
    for N, H, Ci, Co, nf, nd in [LAYERS[i] for i in sel]:
        desc = ops.make_desc(N, H, H, Ci, Co, 1, 1, 1, 0, dt)
        wm = (torch.randn(Co, 1, 1, Ci, generator=g) * 0.05).to(dev)
        wf, wt = ops.pack_weights(wm, Co, 1, Ci, Ci, dt)
        x = ops.nhwc_empty(N, Ci, H, H, dt, dev).normal_()
        dy = ops.nhwc_empty(N, Co, H, H, dt, dev).normal_()
        # COLD=1: rotate over enough operand copies that a launch never finds its inputs in the 256-MB Infinity Cache
        cold = os.environ.get('COLD', '0') == '1'
        nrot = max(2, int(600e6 // (2 * N * H * H * (Ci + Co)) + 1)) if cold else 1
        xs = [x] + [x.clone() for _ in range(nrot - 1)]
        dys = [dy] + [dy.clone() for _ in range(nrot - 1)]
        rot = [0]

        def nxt(lst):
            rot[0] += 1
            return lst[rot[0] % nrot]
        M = N * H * H
        fl = 2.0 * M * Ci * Co
        by = 2.0 * (M * Ci + M * Co + Ci * Co)
        wb = wm.reshape(Co, Ci).to(dt).float()
        # correctness: forward, forward + statistics (values only), input gradient
        y = ops.conv_fwd(desc, x, wf)
        ref = x.permute(0, 2, 3, 1).reshape(M, Ci).float() @ wb.t()
        e_f = float((y.permute(0, 2, 3, 1).reshape(M, Co).float() - ref).abs().max() / (ref.abs().max() + 1e-6))
        ys, part = ops.conv_fwd_stats(desc, x, wf, None)
        e_s = float((ys.float() - y.float()).abs().max())
        dx = ops.conv_dgrad(desc, dy, wt)
        refd = dy.permute(0, 2, 3, 1).reshape(M, Co).float() @ wb
        e_d = float((dx.permute(0, 2, 3, 1).reshape(M, Ci).float() - refd).abs().max() / (refd.abs().max() + 1e-6))
        t_f = timeit(lambda: ops.conv_fwd(desc, nxt(xs), wf))
        t_s = timeit(lambda: ops.conv_fwd_stats(desc, nxt(xs), wf, None))
        t_d = timeit(lambda: ops.conv_dgrad(desc, nxt(dys), wt))
        tot['fwd'] += t_f * nf; tot['fwd_stats'] += t_s * nf; tot['dgrad'] += t_d * nd
        print('%s %4d->%4d @%2d  fwd %6.1f us %5.0f TF/s %5.0f GB/s | +stats %6.1f us | dgrad %6.1f us %5.0f TF/s | err %.1e %.1e stats-vs-plain %.1e'
              % (tag, Ci, Co, H, t_f * 1e6, fl / t_f / 1e12, by / t_f / 1e9, t_s * 1e6, t_d * 1e6, fl / t_d / 1e12, e_f, e_d, e_s), flush=True)
    print('%s per iteration: fwd %.3f ms (with statistics %.3f ms), dgrad %.3f ms' % (tag, tot['fwd'] * 1e3, tot['fwd_stats'] * 1e3, tot['dgrad'] * 1e3))


if __name__ == '__main__':
    main()
