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


def _cover_worker(rank, world, port, q):
    """The real model / optimizers (ResNet-50 layout, parameters only: no kernel runs on the CPU) with rank-dependent
    gradients: drive the overlapped reducer exactly as the gradient hooks of a backward would and record every range."""
    from conftest import PKG  # noqa: F401
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import mi355.da_step as ds
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    torch.manual_seed(0)
    bb = models.resnet50(pretrained=False)
    model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True)
    step, opts, scheds = ds.build_training(model)
    g = torch.Generator().manual_seed(7 + rank)
    for n, p in model.named_parameters():
        if not n.startswith('backbone.fc.'):
            p.grad = torch.randn(p.shape, generator=g)
    for o in opts.values():
        o.ensure_flat()                    # what the first step() does: flat P / G / M buffers, parameters re-aliased
    launched = []
    orig = ds._OverlapReducer._launch
    ds._OverlapReducer._launch = lambda self, G, lo, hi: (launched.append((G.data_ptr(), lo, hi)), orig(self, G, lo, hi))[1]
    report = {}
    for name, keys in (('A', ('f', 'h', 'h_adv', 'h_adv2', 'h_adv3')), ('C', ('f',))):
        for o in opts.values():            # fresh rank-dependent gradients in the flat buffers
            for G in o.flat_grads():
                G.copy_(torch.randn(G.shape, generator=g))
        want = {G.data_ptr(): G.clone() for k in keys for G in opts[k].flat_grads()}
        del launched[:]
        red = ds._OverlapReducer(step, keys)
        for stage in ('neck_out', 'layer4_out', 'layer3_out', 'layer2_out', 'layer1_out', 'pool_out'):   # backward order
            red.stage_done(stage)
        early = len(launched)
        red.finish()
        bufs = {G.data_ptr(): G for k in keys for G in opts[k].flat_grads()}
        cover = {ptr: [] for ptr in bufs}
        for ptr, lo, hi in launched:
            cover[ptr].append((lo, hi))
        exact = all(sorted(r)[0][0] == 0 and sorted(r)[-1][1] == bufs[ptr].numel() and
                    all(a[1] == b[0] for a, b in zip(sorted(r), sorted(r)[1:])) for ptr, r in cover.items())
        # the mean itself: gather the pre-exchange buffers of the other rank
        ok_mean = True
        for ptr, G in bufs.items():
            both = [torch.zeros_like(G) for _ in range(world)]
            dist.all_gather(both, want[ptr])
            ok_mean = ok_mean and torch.allclose(G, sum(both) / world, atol=1e-6)
        report[name] = (exact, ok_mean, early, len(launched), sorted(bufs) == sorted(cover))
    # step B: the plain exchange over exactly the three adversarial heads' buffers
    adv = [G.data_ptr() for k in ('h_adv', 'h_adv2', 'h_adv3') for G in opts[k].flat_grads()]
    report['B'] = sorted(G.data_ptr() for G in step._grads(('h_adv', 'h_adv2', 'h_adv3'))) == sorted(adv)
    # every trainable parameter (backbone.fc excluded) lives in exactly one flat slot of exactly one optimizer
    slots = {}
    for k, o in opts.items():
        for f in o._flat:
            if f is not None:
                for p in f['params']:
                    slots[id(p)] = slots.get(id(p), 0) + 1
    names = {id(p): n for n, p in model.named_parameters()}
    report['slots'] = (all(v == 1 for v in slots.values()),
                       sorted(names[i] for i in names if i not in slots) == ['backbone.fc.bias', 'backbone.fc.weight'])
    q.put((rank, report))
    dist.barrier()
    dist.destroy_process_group()


def test_every_stepped_parameter_is_reduced_exactly_once():
    """World size 2, steps A, B and C: the ranges the overlapped reducer launches (hooks in backward order, then the
    final pass) tile every flat gradient buffer of the optimizers about to step exactly once, the result is the mean
    over ranks, step B exchanges exactly the adversarial heads, and backbone.fc is in no bucket."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_cover_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rep in res:
        for name in ('A', 'C'):
            exact, ok_mean, early, total, same = rep[name]
            assert exact and ok_mean and same, (rank, name, rep[name])
            assert early >= 5 and total > early, (rank, name, early, total)      # pieces launched from the hooks + the tail
        assert rep['B'] is True
        assert rep['slots'] == (True, True), rep['slots']


def _worker_bf16(rank, world, port, q):
    from conftest import PKG  # noqa: F401
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MI355_GRAD_BUCKET_BF16='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import mi355.da_step as ds
    assert ds.BF16_BUCKETS
    g = torch.Generator().manual_seed(7 + rank)
    buf = torch.randn(1000, generator=g)
    mine = buf.clone()
    ds._allreduce_mean([buf])
    w, finish = ds._reduce_mean_(mine, async_op=True)      # the overlapped reducer's form
    w.wait(); finish()
    q.put((rank, buf.numpy().copy(), mine.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_buckets_average_within_bf16_rounding():
    """MI355_GRAD_BUCKET_BF16=1: the fp32 gradient ranges are exchanged as bf16 copies; every rank ends with the same values,
    the mean of the bf16-rounded inputs up to one more bf16 rounding."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    inputs = [torch.randn(1000, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
    ref = sum(t.to(torch.bfloat16).float() for t in inputs) / world
    for _, a, b in res:
        assert (abs(a - res[0][1]) == 0).all() and (abs(b - a) == 0).all()
        assert float(abs(torch.from_numpy(a) - ref).max()) <= 2 ** -7 * float(ref.abs().max())
