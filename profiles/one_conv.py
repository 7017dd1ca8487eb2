import sys, os, torch
sys.path[:0] = ['/root/repo', '/root/repo/domain-adaptative-hand-pose-estimation_amd']
import mi355
from mi355 import ops
dev = torch.device('cuda:0'); mi355.load(); dt = torch.bfloat16
kind = os.environ.get('WHICH', 'fwd')
N, H, Ci, Co, k = 64, 64, 256, 256, 3
desc = ops.make_desc(N, H, H, Ci, Co, k, k, 1, 1, dt)
x = ops.nhwc_empty(N, Ci, H, H, dt, dev).normal_(); dy = ops.nhwc_empty(N, Co, H, H, dt, dev).normal_()
w = torch.randn(Co * k * k * Ci, device=dev).to(dt); dw = torch.empty(Co * k * k * Ci, device=dev)
for _ in range(6):
    if kind == 'fwd': ops.conv_fwd(desc, x, w)
    elif kind == 'dgrad': ops.conv_dgrad(desc, dy, w)
    else: ops.conv_wgrad(desc, x, dy, dw, False)
torch.cuda.synchronize()
