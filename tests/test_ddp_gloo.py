"""Data-parallel gradient exchange on 2 CPU ranks (gloo): the bucket selection of DAStep (which optimizers'
gradients are reduced before which update) and the mean all-reduce.  The HIP kernels cannot run here; the
optimizers are stock torch SGD over CPU tensors, which DAStep accepts as well."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone, self.upsampling = nn.Linear(4, 4), nn.Linear(4, 4)
        self.head, self.head_adv, self.head_adv2, self.head_adv3 = (nn.Linear(4, 2) for _ in range(4))


def _worker(rank, world, port, q):
    import sys
    from conftest import PKG  # noqa: F401  (puts the package source root on sys.path in the spawned process)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mi355.da_step import DAStep, _allreduce_mean
    torch.manual_seed(0)
    m = _Toy()
    mk = lambda ps: torch.optim.SGD(ps, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    opts = dict(f=mk(list(m.backbone.parameters()) + list(m.upsampling.parameters())), h=mk(m.head.parameters()),
                h_adv=mk(m.head_adv.parameters()), h_adv2=mk(m.head_adv2.parameters()), h_adv3=mk(m.head_adv3.parameters()))
    step = DAStep(m, opts, {}, track_accuracy=False)
    g = torch.Generator().manual_seed(100 + rank)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    before = {n: p.grad.clone() for n, p in m.named_parameters()}
    # step B exchanges only the three adversarial heads
    _allreduce_mean(step._grads(('h_adv', 'h_adv2', 'h_adv3')))
    after_b = {n: p.grad.clone() for n, p in m.named_parameters()}
    # step C exchanges backbone + neck
    _allreduce_mean(step._grads(('f',)))
    after_c = {n: p.grad.clone() for n, p in m.named_parameters()}
    opts['f'].step()
    npy = lambda d: {k: v.detach().numpy().copy() for k, v in d.items()}     # by value: the process exits before the parent reads
    q.put((rank, npy(before), npy(after_b), npy(after_c), npy(dict(m.named_parameters()))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean_and_bucket_selection():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    (_, b0, ab0, ac0, p0), (_, b1, ab1, ac1, p1) = [(r[0],) + tuple(tt(d) for d in r[1:]) for r in res]
    for n in b0:
        mean = (b0[n] + b1[n]) / 2
        adv = n.startswith('head_adv')
        fx = n.startswith('backbone') or n.startswith('upsampling')
        # after B: adversarial heads averaged, everything else untouched
        assert torch.allclose(ab0[n], mean if adv else b0[n]) and torch.allclose(ab1[n], mean if adv else b1[n]), n
        # after C: backbone + neck averaged too; the main head never exchanged in B/C
        assert torch.allclose(ac0[n], mean if (adv or fx) else b0[n]), n
        if fx:     # identical parameters on both ranks after the update of the exchanged group
            assert torch.equal(p0[n], p1[n]), n
