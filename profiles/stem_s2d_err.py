import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, 'domain-adaptative-hand-pose-estimation_amd')
import mi355
from mi355 import ops
mi355.load()
dev = 'cuda'
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    N, H, W, Co = 4, 256, 256, 64
    x = torch.randn(N, 3, H, W).to(dt).float()
    w = (torch.randn(Co, 3, 7, 7) / 12).to(dt).float()
    ref = F.conv2d(x.double(), w.double(), None, stride=2, padding=3)
    dy = torch.randn_like(ref).to(dt).double()
    wr = w.double().clone().requires_grad_(True)
    F.conv2d(x.double(), wr, None, stride=2, padding=3).backward(dy)
    gref = wr.grad.permute(0, 2, 3, 1)
    wm = w.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = ops.to_nhwc(dy.float().to(dev), dt)
    # folded
    xs = ops.to_nhwc_s2d(x.to(dev), dt)
    d1 = ops.make_desc(N, H // 2, W // 2, 16, Co, 4, 4, 1, 2, dt, out_hw=(H // 2, W // 2))
    y1 = ops.conv_fwd(d1, xs, ops.stem_s2d_pack(wm, dt), None)
    gs = torch.empty(Co * 256, device=dev); ops.conv_wgrad(d1, xs, dyd, gs, False)
    g1 = torch.empty(Co, 7, 7, 3, device=dev); ops.stem_s2d_unpack_grad(gs, g1, False)
    # padded 7x7
    per = 8 if dt == torch.bfloat16 else 4
    xp = ops.to_nhwc(x.to(dev), dt, per)
    d0 = ops.make_desc(N, H, W, per, Co, 7, 7, 2, 3, dt)
    wf, _ = ops.pack_weights(wm, Co, 49, 3, per, dt)
    y0 = ops.conv_fwd(d0, xp, wf, None)
    g0 = torch.empty(Co, 7, 7, per, device=dev); ops.conv_wgrad(d0, xp, dyd, g0, False)
    sc = float(ref.abs().max()); gsc = float(gref.abs().max())
    print(dt, 'fwd err/scale  7x7 %.3e  folded %.3e | wgrad err/scale 7x7 %.3e folded %.3e' % (
        float((y0.double().cpu() - ref).abs().max()) / sc, float((y1.double().cpu() - ref).abs().max()) / sc,
        float((g0[..., :3].double().cpu() - gref).abs().max()) / gsc, float((g1.double().cpu() - gref).abs().max()) / gsc))
