#!/usr/bin/env python
"""Domain-adaptive hand-pose training on the MI355X kernels — same command line, log / checkpoint layout and
training schedule as the reference's ``train1.py`` (main :37-275, pretrain :278-325, train :328-492,
validate :495-536, CLI :591-675).  Additive flags: ``--synthetic`` (seeded synthetic data instead of the
out-of-scope CPU dataset layer), ``--dtype {bf16,f32}``, ``--no-graph``.

    python train1.py data/H3D -t Hand3DStudio --synthetic -a resnet50 -b 64

Data parallel (the reference is single-process): start one process per GPU with torchrun,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train1.py -- data/H3D ... -b 64

(the ``--`` keeps torchrun's own parser away from the script's flags: ``--log`` is an ambiguous prefix of its ``--log-dir``).

``-b`` stays the per-GPU batch (the path shards by image, BatchNorm statistics stay per GPU as in the reference).  Every
rank reads ``RANK / LOCAL_RANK / WORLD_SIZE`` before touching the GPU, draws its own shard of both training sets
(``DistributedSampler``), the gradient mean of the optimizers about to step is exchanged over RCCL (``mi355.da_step``),
validation counts are summed over ranks, and only rank 0 writes the log and the checkpoints (reference sites made
rank-aware: train1.py:54-99 loaders, :141-154 optimizers, :248-268 checkpoints).
"""
import argparse
import os
import random
import shutil
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import LambdaLR, MultiStepLR
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

import mi355
import uda.model as models
from mi355 import ops as _ops
from mi355.da_step import build_training, broadcast_module, _allreduce_mean
from mi355.optim import FusedSGD
from uda.model.loss import JointsKLLoss
from uda.model.pose_resnet2 import Upsampling, PoseResNet
from uda.model.regda_7 import PoseResNetx9 as RegDAPoseResNetx1, PoseResNetx10 as RegDAPoseResNetx2
from utils.data import ForeverDataIterator, DevicePrefetcher
from utils.keypoint_detection import accuracy
from utils.logger import CompleteLogger
from utils.meter import AverageMeter, ProgressMeter, AverageMeterDict

device = torch.device("cuda" if torch.cuda.device_count() > 0 else "cpu")     # device_count() does not initialise the GPU
RANK, WORLD = 0, 1


def init_distributed():
    """One process per GPU: read the torchrun environment BEFORE any GPU call, bind this process to its GPU and join
    the process group (backend nccl = RCCL over xGMI; MI355_DIST_BACKEND=gloo for rehearsals with several ranks on
    one GPU).  Single-process runs (no WORLD_SIZE) skip all of it."""
    global device, RANK, WORLD
    WORLD = int(os.environ.get('WORLD_SIZE', '1'))
    RANK = int(os.environ.get('RANK', '0'))
    if WORLD > 1:
        local = int(os.environ.get('LOCAL_RANK', str(RANK)))
        ndev = torch.cuda.device_count()
        if ndev == 0:
            raise SystemExit('this training path needs an MI355X (HIP kernels only, no CPU fallback)')
        torch.cuda.set_device(local % ndev)
        device = torch.device('cuda', local % ndev)
        if WORLD > ndev:
            # several ranks share a GPU (rehearsal): the one-launch BatchNorm backward wants the whole chip for its resident
            # blocks -- two processes launching it at once would starve each other (include/mi355pose.h, mi355_bn_set_resident)
            os.environ['MI355_BN_RESIDENT'] = '0'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('MI355_DIST_BACKEND', 'nccl'), rank=RANK, world_size=WORLD)
    return RANK, WORLD


def replicas_in_sync(model):
    """Failure detection for the data-parallel run: every rank must hold bit-identical parameters (same initial
    broadcast, same averaged gradients, same deterministic update kernels).  Returns the checksum, raises on drift."""
    cs = torch.stack([p.detach().double().abs().sum() for p in model.parameters()]).sum().reshape(1)
    if WORLD > 1:
        every = [torch.zeros_like(cs) for _ in range(WORLD)]
        dist.all_gather(every, cs)
        if any(float(e) != float(every[0]) for e in every):
            raise RuntimeError('data-parallel replicas diverged: parameter checksums %s' % [float(e) for e in every])
    return float(cs)


def build_datasets(args):
    image_size, heatmap_size = (args.image_size,) * 2, (args.heatmap_size,) * 2
    if args.synthetic:
        from utils.synthetic_dataset import SyntheticHand21
        mk = lambda n, seed: SyntheticHand21(n, image_size, heatmap_size, seed=seed)
        n = args.batch_size * max(args.iters_per_epoch, 1)
        return mk(n, 11), mk(4 * args.batch_size, 12), mk(n, 13), mk(4 * args.batch_size, 14)
    import uda.dataset as datasets                   # RHD / H3D / STB readers + key-point aware augmentation (PIL + numpy)
    import uda.dataset.keypoint_detection as T
    normalize = T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    train_tf = T.Compose([T.RandomRotation(args.rotation), T.RandomResizedCrop(size=args.image_size, scale=args.resize_scale),
                          T.ColorJitter(brightness=0.25, contrast=0.25, saturation=0.25), T.GaussianBlur(), T.ToTensor(), normalize])
    val_tf = T.Compose([T.Resize(args.image_size), T.ToTensor(), normalize])
    src, tgt = datasets.__dict__[args.source], datasets.__dict__[args.target]
    kw = dict(image_size=image_size, heatmap_size=heatmap_size)
    return (src(root=args.source_root, transforms=train_tf, **kw), src(root=args.source_root, split='test', transforms=val_tf, **kw),
            tgt(root=args.target_root, transforms=train_tf, **kw), tgt(root=args.target_root, split='test', transforms=val_tf, **kw))


def make_loader(ds, args, train):
    """train: this rank's shard (reshuffled every pass, ForeverDataIterator advances the sampler epoch); validation: the
    strided shard rank::WORLD without padding, so that the counts summed over ranks are exactly the data set's."""
    if WORLD == 1:
        return DataLoader(ds, batch_size=args.batch_size, shuffle=train, num_workers=args.workers if train else 0,
                          pin_memory=True, drop_last=train)
    if train:
        sampler = DistributedSampler(ds, num_replicas=WORLD, rank=RANK, shuffle=True, seed=args.seed or 0, drop_last=True)
        return DataLoader(ds, batch_size=args.batch_size, sampler=sampler, num_workers=args.workers, pin_memory=True, drop_last=True)
    return DataLoader(ds, batch_size=args.batch_size, sampler=list(range(RANK, len(ds), WORLD)), num_workers=0, pin_memory=True)


def main(args):
    init_distributed()
    logger = CompleteLogger(args.log, args.phase, quiet=RANK != 0)      # rank 0 owns the console mirror and the log file
    print(args)
    if device.type != 'cuda':
        raise SystemExit('this training path needs an MI355X (HIP kernels only, no CPU fallback)')
    mi355.load()
    mi355.set_compute_dtype(args.dtype)
    if args.seed is not None:
        random.seed(args.seed + RANK)
        torch.manual_seed(args.seed)             # same initial weights everywhere (and broadcast below anyway)
        warnings.warn('You have chosen to seed training.')
    if WORLD > 1:
        print('data parallel: %d ranks, backend %s, per-GPU batch %d' % (WORLD, dist.get_backend(), args.batch_size))

    train_s, val_s, train_t, val_t = build_datasets(args)
    ld = lambda ds, train: make_loader(ds, args, train)
    train_source_loader, val_source_loader = ld(train_s, True), ld(val_s, False)
    train_target_loader, val_target_loader = ld(train_t, True), ld(val_t, False)
    print("Source train:", len(train_source_loader)); print("Target train:", len(train_target_loader))
    print("Source test:", len(val_source_loader)); print("Target test:", len(val_target_loader))
    train_source_iter, train_target_iter = ForeverDataIterator(train_source_loader), ForeverDataIterator(train_target_loader)
    # host -> HBM copies of the next batch overlap the current step (pinned, double-buffered, side stream)
    train_source_iter, train_target_iter = DevicePrefetcher(train_source_iter, device), DevicePrefetcher(train_target_iter, device)

    # model (+ the frozen EMA copy the reference builds and checkpoints, train1.py:102-128)
    backbone = models.__dict__[args.arch](pretrained=True)
    upsampling = Upsampling(backbone.out_features)
    num_keypoints = train_s.num_keypoints
    model = RegDAPoseResNetx1(backbone, upsampling, 256, num_keypoints, num_head_layers=args.num_head_layers, finetune=True).to(device)
    ema_bb = models.__dict__[args.arch](pretrained=False)
    model_ema = RegDAPoseResNetx2(ema_bb, Upsampling(ema_bb.out_features), 256, num_keypoints,
                                  num_head_layers=args.num_head_layers, finetune=True).to(device)
    for p_main, p_ema in zip(model.parameters(), model_ema.parameters()):
        p_ema.data.copy_(p_main.data)
        p_ema.requires_grad = False

    criterion = JointsKLLoss()
    step, opts, scheds = build_training(model, heatmap_size=args.heatmap_size, lr=args.lr, momentum=args.momentum, wd=args.wd,
                                        lr_gamma=args.lr_gamma, lr_decay=args.lr_decay, trade_off=args.trade_off,
                                        num_keypoints=num_keypoints)
    if args.synthetic:
        # noise images make the target predictions collapse within a few dozen iterations; the reference's per-map
        # max-normalisation then divides 0 by 0 (regda_7.py:3623-3625).  Synthetic runs keep such maps at zero instead.
        for c in step.crit.values():
            if hasattr(c, 'guard_empty_maps'):
                c.guard_empty_maps = True
    start_epoch = 0
    if args.resume is None:
        if args.pretrain is None or (args.synthetic and not os.path.exists(args.pretrain)):
            print("Pretraining the model on source domain.")
            args.pretrain = logger.get_checkpoint_path('pretrain')
            pre = PoseResNet(backbone, upsampling, 256, num_keypoints, True).to(device)
            optimizer = FusedSGD(pre.get_parameters(lr=args.lr), lr=args.lr, momentum=args.momentum, weight_decay=args.wd, nesterov=True)
            lr_scheduler = MultiStepLR(optimizer, args.lr_step, args.lr_factor)
            best_acc = -1
            for epoch in range(args.pretrain_epochs):
                lr_scheduler.step()                      # the reference steps the schedule before the epoch (train1.py:167)
                pretrain(train_source_iter, pre, criterion, optimizer, epoch, args)
                acc = validate(val_source_loader, pre, criterion, args)
                _ops.bn_resident_check('pre-training epoch %d' % epoch)   # (validation has synchronised: the poll is free)
                if acc['all'] > best_acc:
                    best_acc = acc['all']
                    if RANK == 0:
                        torch.save({'model': pre.state_dict()}, args.pretrain)
                print("Source: {} best: {}".format(acc['all'], best_acc))
            if WORLD > 1:
                dist.barrier()                           # rank 0 has written the file everybody reads next
        pretrained_dict = torch.load(args.pretrain, map_location='cpu', weights_only=False)['model']
        model_dict = model.state_dict()
        pretrained_dict = {k: v for k, v in pretrained_dict.items() if k in model_dict}
        model.load_state_dict(pretrained_dict, strict=False)
        model_ema.load_state_dict(pretrained_dict, strict=False)
    else:
        ck = torch.load(args.resume, map_location='cpu', weights_only=False)
        model.load_state_dict(ck['model']); model_ema.load_state_dict(ck['model'])
        for k, name in (('f', 'optimizer_f'), ('h', 'optimizer_h'), ('h_adv', 'optimizer_h_adv')):
            opts[k].load_state_dict(ck[name]); scheds[k].load_state_dict(ck['lr_scheduler' + name[len('optimizer'):]])
        for k in ('h_adv2', 'h_adv3'):                    # additive keys (the reference never saved these two)
            if 'optimizer_' + k in ck:
                opts[k].load_state_dict(ck['optimizer_' + k]); scheds[k].load_state_dict(ck['lr_scheduler_' + k])
        model.gl_layer.iter_num = ck.get('gl_iter_num', 0)
        start_epoch = ck['epoch'] + 1
    broadcast_module(model)                              # replicas start bit-identical whatever each rank loaded
    broadcast_module(model_ema)

    if args.phase == 'test':
        s_acc = validate(val_source_loader, model, criterion, args)
        t_acc = validate(val_target_loader, model, criterion, args)
        print("Source: {:4.3f} Target: {:4.3f}".format(s_acc['all'], t_acc['all']))
        for name, acc in t_acc.items():
            print("{}: {:4.3f}".format(name, acc))
        logger.close()
        return

    best_acc = 0
    print("Start regression domain adaptation.")
    for epoch in range(start_epoch, args.epochs):
        logger.set_epoch(epoch)
        print(*[scheds[k].get_last_lr() for k in ('f', 'h', 'h_adv', 'h_adv2')])
        train(train_source_iter, train_target_iter, step, scheds, epoch, args)
        s_acc = validate(val_source_loader, model, criterion, args)
        t_acc = validate(val_target_loader, model, criterion, args)
        # a one-launch BatchNorm backward whose blocks could not all get onto the chip has written NaN gradients: raise
        # instead of training on (the poll synchronises, validation just has)
        step.check_health('epoch %d' % epoch)
        if WORLD > 1:
            print('replicas in sync (parameter checksum %.6e)' % replicas_in_sync(model))
        if RANK != 0:
            if WORLD > 1:
                dist.barrier()
            best_acc = max(best_acc, t_acc['all'])
            continue
        torch.save({'model': model.state_dict(),
                    'optimizer_f': opts['f'].state_dict(), 'optimizer_h': opts['h'].state_dict(),
                    'optimizer_h_adv': opts['h_adv'].state_dict(),
                    'lr_scheduler_f': scheds['f'].state_dict(), 'lr_scheduler_h': scheds['h'].state_dict(),
                    'lr_scheduler_h_adv': scheds['h_adv'].state_dict(), 'epoch': epoch, 'args': args,
                    # additive keys: close the reference's resume gaps
                    'optimizer_h_adv2': opts['h_adv2'].state_dict(), 'optimizer_h_adv3': opts['h_adv3'].state_dict(),
                    'lr_scheduler_h_adv2': scheds['h_adv2'].state_dict(), 'lr_scheduler_h_adv3': scheds['h_adv3'].state_dict(),
                    'gl_iter_num': model.gl_layer.iter_num}, logger.get_checkpoint_path(epoch))
        torch.save({'model_ema': model_ema.state_dict()}, logger.get_checkpoint_path('model_ema'))
        if t_acc['all'] > best_acc:
            shutil.copy(logger.get_checkpoint_path(epoch), logger.get_checkpoint_path('best'))
            best_acc = t_acc['all']
        if WORLD > 1:
            dist.barrier()                               # checkpoints of this epoch are complete
        print("Source: {:4.3f} Target: {:4.3f} Target(best): {:4.3f}".format(s_acc['all'], t_acc['all'], best_acc))
        for name, acc in t_acc.items():
            print("{}: {:4.3f}".format(name, acc))
    logger.close()
    if WORLD > 1:
        dist.destroy_process_group()


def pretrain(train_source_iter, model, criterion, optimizer, epoch, args):
    batch_time, data_time = AverageMeter('Time', ':4.2f'), AverageMeter('Data', ':3.1f')
    losses_s, acc_s = AverageMeter('Loss (s)', ":.2e"), AverageMeter("Acc (s)", ":3.2f")
    progress = ProgressMeter(args.iters_per_epoch, [batch_time, data_time, losses_s, acc_s], prefix="Epoch: [{}]".format(epoch))
    model.train()
    end = time.time()
    for i in range(args.iters_per_epoch):
        optimizer.zero_grad()
        x_s, label_s, weight_s, _ = next(train_source_iter)
        x_s, label_s, weight_s = x_s.to(device, non_blocking=True), label_s.to(device, non_blocking=True), weight_s.to(device, non_blocking=True)
        data_time.update(time.time() - end)
        y_s = model(x_s)
        loss_s = criterion(y_s, label_s, weight_s)
        loss_s.backward()
        if WORLD > 1:
            _allreduce_mean(optimizer.flat_grads())     # gradient mean over ranks (the flat buffers are the buckets)
        optimizer.step()
        if i % args.print_freq == 0:                    # host reads only when something is printed
            _, avg_acc_s, cnt_s, _ = accuracy(y_s.detach(), label_s)
            acc_s.update(avg_acc_s, cnt_s); losses_s.update(float(loss_s), cnt_s)
            batch_time.update(time.time() - end)
            progress.display(i)
        end = time.time()


def _pck(dists, thr=0.5):
    """avg accuracy + count from a (B,K) device tensor of PCK distances (-1 = ignored), utils/keypoint_detection.py:53-92."""
    d = dists.t().cpu().numpy()
    accs = [float((row[row != -1] < thr).mean()) for row in d if (row != -1).any()]
    return (sum(accs) / len(accs) if accs else 0), len(accs)


def train(train_source_iter, train_target_iter, step, scheds, epoch, args):
    names = ['Time', 'Data', 'Loss (s)', 'Loss (t, false)', 'Loss (t, truth)', 'Acc (s)', 'Acc (t)', 'Acc (s, adv)', 'Acc (t, adv)']
    fmts = [':4.2f', ':3.1f', ':.2e', ':.2e', ':.2e', ':3.2f', ':3.2f', ':3.2f', ':3.2f']
    meters = [AverageMeter(n, f) for n, f in zip(names, fmts)]
    progress = ProgressMeter(args.iters_per_epoch, meters, prefix="Epoch: [{}]".format(epoch))
    end = time.time()
    for i in range(args.iters_per_epoch):
        x_s, label_s, weight_s, _ = next(train_source_iter)
        x_t, label_t, weight_t, _ = next(train_target_iter)
        to = lambda t: t.to(device, non_blocking=True)
        batch = dict(x_s=to(x_s), label_s=to(label_s), w_s=to(weight_s), x_t=to(x_t), w_t=to(weight_t), label_t=to(label_t))
        meters[1].update(time.time() - end)
        # HIP-graph replay once this process has run three eager iterations (whatever epoch it resumed at).  With several
        # ranks the eager path overlaps the gradient exchange with the backward, which only pays while the host can enqueue an
        # iteration faster than the GPU runs it: the fourth iteration is timed on host and GPU and every rank takes the same
        # decision (DAStep.choose_launch_mode, the rule bench.py uses; MI355_DDP_GRAPH=1 / 0 forces it)
        step.eager_iters = getattr(step, 'eager_iters', 0)
        if step.graphs is None and not args.no_graph and step.eager_iters == 3 and not getattr(step, 'mode_chosen', False):
            step.mode_chosen = True
            mode = 'graph'
            if WORLD > 1:
                mode = step.choose_launch_mode(batch, after=lambda: [s.step() for s in scheds.values()])
                print('multi-rank launch mode: host / GPU time of an eager iteration %.2f -> %s'
                      % (step.host_gpu_ratio, ('HIP-graph replay, overlapped RCCL gradient exchange captured with the graphs' if step._overlap_capturable()
                                                 else 'HIP-graph replay, collectives between the graphs') if mode == 'graph'
                         else 'eager launches, gradient exchange overlapped with the backward'))
                step.eager_iters += 1
                end = time.time()
                if mode == 'graph':
                    step.capture(batch, warmup=0)
                continue                                      # (the timed iteration was a real one)
            step.capture(batch, warmup=0)
            print('HIP graphs captured: iterations replay six graphs from here on')
        elif step.graphs is None and step.eager_iters == 0:
            print('eager kernel launches%s' % (' with the gradient exchange overlapped with the backward' if WORLD > 1 else ''))
        step.eager_iters += 1
        out = step.run(batch)                            # steps A, B, C + GL step (train1.py:371-453)
        for s in scheds.values():
            s.step()
        if i % args.print_freq == 0:
            for m, k in zip(meters[2:5], ('loss_s', 'loss_gf', 'loss_gt')):
                m.update(float(out[k]), args.batch_size)
            for m, k in zip(meters[5:], ('pck_s', 'pck_t', 'pck_s_adv', 'pck_t_adv')):
                a, c = _pck(out[k]); m.update(a, c)
            meters[0].update(time.time() - end)
            progress.display(i)
        end = time.time()


def validate(val_loader, model, criterion, args):
    batch_time, losses = AverageMeter('Time', ':6.3f'), AverageMeter('Loss', ':.2e')
    acc = AverageMeterDict(val_loader.dataset.keypoints_group.keys(), ":3.2f")
    progress = ProgressMeter(len(val_loader), [batch_time, losses, acc['all']], prefix='Test: ')
    model.eval()
    from mi355.infer import GraphedForward
    forward = GraphedForward(model)          # full batches replay one HIP graph; the ragged last batch runs eagerly
    with torch.no_grad():
        end = time.time()
        for i, (x, label, weight, meta) in enumerate(val_loader):
            x, label, weight = x.to(device), label.to(device), weight.to(device)
            y = forward(x)
            loss = criterion(y, label, weight)
            losses.update(loss.item(), x.size(0))
            acc_per_points, avg_acc, cnt, pred = accuracy(y, label)
            acc.update(val_loader.dataset.group_accuracy(acc_per_points), x.size(0))
            batch_time.update(time.time() - end)
            end = time.time()
            if i % args.print_freq == 0:
                progress.display(i)
    if WORLD > 1:                                        # sums and counts over all shards (train1.py:505-524 per shard)
        keys = list(acc.dict.keys())
        t = torch.tensor([v for k in keys for v in (acc[k].sum, acc[k].count)] + [losses.sum, losses.count],
                         dtype=torch.float64, device=device)
        dist.all_reduce(t)
        t = t.cpu().tolist()
        for j, k in enumerate(keys):
            acc[k].sum, acc[k].count = t[2 * j], t[2 * j + 1]
            acc[k].avg = acc[k].sum / max(acc[k].count, 1)
    return acc.average()


# (flags, kwargs) for every option of the reference's command line (train1.py:602-674: same names, types and
# defaults), followed by the additive ones of this implementation
_OPTIONS = [
    (('--source_root',), dict(default='data/RHD', help='source dataset directory')),
    (('target_root',), dict(help='target dataset directory')),
    (('-s', '--source'), dict(default='RenderedHandPose', help='source dataset class')),
    (('-t', '--target'), dict(help='target dataset class')),
    (('--resize-scale',), dict(nargs='+', type=float, default=(0.6, 1.3), help='RandomResizedCrop scale range')),
    (('--rotation',), dict(type=int, default=180, help='RandomRotation range in degrees')),
    (('--image-size',), dict(type=int, default=256, help='network input side')),
    (('--heatmap-size',), dict(type=int, default=64, help='heat-map side of the main head')),
    (('-a2', '--arch2'), dict(metavar='ARCH', default='net_hg', help='unused (reference CLI parity)')),
    (('--pretrain',), dict(type=str, default='models/pretrain_rhd.pth', help='source-only pre-training checkpoint')),
    (('--ema_model',), dict(type=str, default=None, help='unused (reference CLI parity)')),
    (('--resume',), dict(type=str, default=None, help='checkpoint to continue from')),
    (('--resume2',), dict(type=str, default=None, help='unused (reference CLI parity)')),
    (('--num-head-layers',), dict(type=int, default=2)),
    (('--margin',), dict(type=float, default=4., help='unused (reference CLI parity)')),
    (('--trade-off',), dict(default=1., type=float, help='weight of the target-domain disparity losses')),
    (('-b', '--batch-size'), dict(default=32, type=int, metavar='N', help='images per domain per iteration')),
    (('--lr', '--learning-rate'), dict(default=0.01, type=float, metavar='LR', dest='lr', help='base learning rate')),
    (('--momentum',), dict(default=0.9, type=float, metavar='M')),
    (('--wd', '--weight-decay'), dict(default=0.0001, type=float, metavar='W')),
    (('--lr-gamma',), dict(default=0.0001, type=float, help='inverse-decay schedule: lr * (1 + gamma * it) ** -decay')),
    (('--lr-decay',), dict(default=0.75, type=float)),
    (('--lr-step',), dict(default=[45, 60], type=tuple, help='pre-training MultiStepLR milestones')),
    (('--lr-factor',), dict(default=0.1, type=float, help='pre-training MultiStepLR factor')),
    (('-j', '--workers'), dict(default=4, type=int, metavar='N', help='loader worker processes')),
    (('--pretrain_epochs',), dict(default=70, type=int, metavar='N')),
    (('--epochs',), dict(default=200, type=int, metavar='N')),
    (('-i', '--iters-per-epoch'), dict(default=500, type=int)),
    (('-p', '--print-freq'), dict(default=100, type=int, metavar='N')),
    (('--seed',), dict(default=1, type=int)),
    (('--log',), dict(type=str, default='logs/mt', help='run directory (logs, checkpoints, images)')),
    (('--phase',), dict(type=str, default='train', choices=['train', 'test'])),
    (('--debug',), dict(action='store_true', help='accepted for CLI parity (visualisation is out of scope)')),
    (('--ema-decay',), dict(default=0.999, type=float, metavar='ALPHA', help='unused (reference CLI parity)')),
    # additive
    (('--synthetic',), dict(action='store_true', help='seeded synthetic batches instead of the CPU dataset layer')),
    (('--dtype',), dict(default='bf16', choices=['bf16', 'f32', 'fp8'], help="compute dtype of activations / packed weights ('fp8': bf16 storage, fp8 operands in the K-heavy conv GEMMs)")),
    (('--no-graph',), dict(action='store_true', help='launch kernels eagerly instead of replaying HIP graphs')),
]


def build_parser(description='Domain-adaptive hand-pose training on MI355X'):
    archs = sorted(n for n in models.__dict__ if n.islower() and not n.startswith('__') and callable(models.__dict__[n]))
    parser = argparse.ArgumentParser(description=description)
    parser.add_argument('-a', '--arch', metavar='ARCH', default='resnet101', choices=archs, help=' | '.join(archs))
    for flags, kw in _OPTIONS:
        parser.add_argument(*flags, **kw)
    return parser


if __name__ == '__main__':
    main(build_parser().parse_args())
