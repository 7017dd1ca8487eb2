"""Two data-parallel ranks sharing the one GPU of the test box (gloo collectives on device tensors): the gradient exchange
that is started piecewise from gradient hooks while the backward is still running must give exactly the parameters of the
plain exchange after the backward, identical on both ranks.  (RCCL needs one GPU per rank; the driver's 8-GPU run covers it.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, overlap, q):
    from conftest import PKG  # noqa: F401  (package source root on sys.path in the spawned process)
    import torch.distributed as dist
    # (same BatchNorm-backward kernels in both runs: the overlapped pass switches the one-launch form off, da_step._begin_reduce)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), MI355_OVERLAP_ALLREDUCE='1' if overlap else '0',
                      MI355_BN_RESIDENT='0')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import mi355
    import mi355.da_step as ds
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from utils.synthetic import make_batch
    mi355.load(); mi355.set_compute_dtype('f32')
    torch.manual_seed(1)
    bb = models.resnet18(pretrained=False)
    model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
    ds.broadcast_module(model, 0)
    step, opts, scheds = ds.build_training(model, heatmap_size=32)
    batch = make_batch(2, 128, 32, seed=1 + rank, device=dev)
    pieces = []
    orig = ds._OverlapReducer._launch
    ds._OverlapReducer._launch = lambda self, G, lo, hi: (pieces.append((lo, hi)), orig(self, G, lo, hi))[1]
    for _ in range(3):
        step.run(batch)
        for s in scheds.values():
            s.step()
    torch.cuda.synchronize()
    cs = [float(p.detach().double().abs().sum()) for p in model.parameters()]
    q.put((rank, cs, len(pieces)))
    dist.barrier()
    dist.destroy_process_group()


def _run(overlap):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_overlapped_gradient_exchange_equals_plain_exchange(gpu):
    on, off = _run(True), _run(False)
    assert on[0][1] == on[1][1], 'ranks diverged with the overlapped exchange'
    assert off[0][1] == off[1][1]
    assert on[0][1] == off[0][1], 'overlapped and plain exchange disagree'
    assert on[0][2] > 10 and off[0][2] == 0          # the pieces really were launched from the hooks


def test_train_cli_with_two_ranks(gpu, tmp_path):
    """train1.py under torchrun with 2 ranks (gloo on the one test GPU): per-rank data shards, gradient exchange in the
    pre-training and the A/B/C loop, summed validation counts, rank-0-only log / checkpoints, replicas bit-identical
    after the epoch (the script checks parameter checksums over ranks and raises on drift); test.py on the result."""
    import subprocess
    import sys
    from conftest import PKG
    log = str(tmp_path / 'run')
    env = dict(os.environ, PYTHONPATH=PKG, MI355_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    launch = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
              '--master-port', str(_free_port())]
    common = ['data/none', '-t', 'Hand3DStudio', '--synthetic', '-a', 'resnet18', '-b', '4', '-i', '5', '-p', '2', '-j', '0',
              '--image-size', '128', '--heatmap-size', '32', '--pretrain_epochs', '1', '--log', log]
    r = subprocess.run(launch + [os.path.join(PKG, 'train1.py'), '--'] + common + ['--epochs', '1', '--pretrain', str(tmp_path / 'none.pth')],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = r.stdout
    assert 'data parallel: 2 ranks' in out and 'replicas in sync' in out and 'Target(best)' in out
    assert out.count('Start regression domain adaptation.') == 1          # rank 0 only
    assert 'gradient exchange overlapped with the backward' in out
    ck_path = os.path.join(log, 'checkpoints', '0.pth')
    assert os.path.exists(ck_path) and os.path.exists(os.path.join(log, 'checkpoints', 'pretrain.pth'))
    logs = [f for f in os.listdir(log) if f.endswith('.txt')]
    assert len(logs) == 1, logs
    # validation counts are summed over the two shards: 4*B = 16 samples per split
    r = subprocess.run(launch[:-1] + [str(_free_port()), os.path.join(PKG, 'test.py'), '--'] + common + ['--checkpoint', ck_path],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'Source:' in r.stdout and 'fingertip:' in r.stdout


def test_rccl_single_rank_smoke(gpu):
    """RCCL itself (backend 'nccl'), which the two-rank tests above cannot use on a one-GPU box: a ONE-rank group with every
    collective of the multi-rank path forced on (MI355_DDP_FORCE_COLLECTIVES) -- init, parameter broadcast of the conv-form
    views, asynchronous AVG all-reduces of flat fp32 gradient ranges started from the backward's hooks, the same exchange CAPTURED
    into the replayed graphs (and, switched off, the blocking exchange between the graphs), the MAX all-reduce of the launch-mode
    decision, barrier (tests/rccl_single_rank.py, one process per
    run).  A mean over one rank is the identity: the parameters after six iterations must equal those of the same run without
    a process group, bit for bit."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = []
    for use_dist, graph_overlap in (('1', '1'), ('1', '0'), ('0', '1')):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), MI355_DDP_FORCE_COLLECTIVES=use_dist,
                   MI355_OVERLAP_ALLREDUCE='1', MI355_BN_RESIDENT='0',        # (same BatchNorm-backward kernels in all runs)
                   MI355_DDP_GRAPH_OVERLAP=graph_overlap)
        r = subprocess.run([sys.executable, os.path.join(here, 'rccl_single_rank.py'), use_dist], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-3000:]
        out.append(json.loads(r.stdout.strip().splitlines()[-1]))
    d, b, p = out
    assert d['collective_launches'] > 0 and p['collective_launches'] == 0      # the overlapped reducer really issued RCCL collectives
    assert d['mode'] in ('graph', 'eager') and d['backend'] == 'nccl'
    # graph replay: the overlapped exchange was captured into the graphs (its all-reduces issued during capture, none from the host
    # at replay) -- or, switched off, stayed between the graphs; either way the same bits as without a process group
    assert d['exchange_captured'] and d['captured_launches'] > 0
    assert not b['exchange_captured'] and b['captured_launches'] == 0
    assert d['checksums'] == p['checksums'] and b['checksums'] == p['checksums']
