#!/usr/bin/env python
"""bench.py — throughput of the domain-adaptation training iteration (steps A+B+C of train1.py:371-458)
on synthetic 256x256 batches, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full iteration: 3 forward + 3 backward passes over B images each (B source + B target images),
5 SGD updates, pseudo-labels / losses / PCK bookkeeping on the device.  value = unique images / s over the
whole job = 2*B*N / t_iter.  Prints ONE JSON line on rank 0 (contract in the task statement):
metric/value/unit, roofline (MFMA conv family, timed live with HIP events in a second pass over the same
steps) and cpu_baseline (the CPU oracle = op-for-op restatement of the reference, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'domain-adaptative-hand-pose-estimation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# forward conv+deconv GFLOPs per image (2*MAC), SURVEY.md table A3 / BASELINE.md section 4
F_GFLOP = {('resnet18', 128): 5.611, ('resnet18', 256): 22.443, ('resnet50', 256): 29.188,
           ('resnet101', 256): 38.885, ('resnet101', 512): 155.540}
# dense MFMA peaks, MI355X_MICROARCH.md.  'fp8' lines are priced against the chip's dense fp8 peak (5 PFLOP/s): the fp8 kernels use
# the K = 64 instruction v_mfma_f32_32x32x64_f8f6f4 (the block-scaled opcode's encoding with unit scales), which issues at that rate.
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3, 'fp8': 5000.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--arch', default='resnet50')
    ap.add_argument('--batch-size', type=int, default=64, help='per-GPU batch (source) = per-GPU batch (target)')
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'fp8'],
                    help="'fp8': bf16 storage, fp8 (e4m3 / e5m2) operands in the forward / input-gradient GEMMs of the K-heavy convs")
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying HIP graphs')
    ap.add_argument('--graph', action='store_true', help='replay HIP graphs also with more than one rank (default there: eager; '
                    'at B=64 the iteration is GPU-bound either way: 44.72 ms replayed vs 44.75 ms eager)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--launch-log', default=None, metavar='PATH', help='after the roofline pass run ONE more eager iteration with every logged '
                    'kernel family bracketed by events (conv MFMA, BatchNorm, slab reductions) and write the per-launch list as JSON: '
                    'the input of profiles/insitu_table.py')
    ap.add_argument('--no-eval', action='store_true', help='skip the forward-only (test.py path) section: profiler runs that should see training iterations only')
    ap.add_argument('--cpu-baseline-batch', type=int, default=4)
    ap.add_argument('--cpu-baseline-iters', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL over xGMI (default); gloo only to rehearse the '
                    'multi-process path on one GPU (all ranks on device 0)')
    return ap.parse_args()


def log(*a):
    print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def pmc_record(args):
    """Counter evidence collected offline on this same workload (profiles/collect_pmc.sh -> profiles/rNN_pmc_traffic.json:
    rocprofv3 kernel trace + separate --pmc FETCH_SIZE / WRITE_SIZE passes, folded per kernel family by
    profiles/pmc_fold.py with the gfx950 correction of MI355X_MICROARCH.md).  Returns the record of this configuration
    (or None) and the name + git commit of the file, so that a stale file is visible in the JSON line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic.json')))      # the newest round's file
    try:
        path = files[-1]
        rec = json.load(open(path))
        key = '%s_%d_b%d_%s' % (args.arch, args.image_size, args.batch_size, args.dtype)
        cfg = rec['configs'][key]
        return cfg, {'file': 'profiles/' + os.path.basename(path), 'collected_at_commit': cfg.get('git_head'), 'command': cfg.get('command')}
    except Exception:
        return None, None


def cpu_baseline(arch, image_size, batch, iters):
    """The CPU oracle (pure-torch fp32 restatement of the reference iteration) on the host cores."""
    from oracle.train_step import build_model, DATrainer
    from utils.synthetic import make_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    log('cpu baseline on %d threads' % cores)
    torch.manual_seed(1)
    model = build_model(arch)
    tr = DATrainer(model, heatmap_size=image_size // 4)
    b = make_batch(batch, image_size, image_size // 4, seed=1, device='cpu', with_target_labels=False)
    tr.step(b['x_s'], b['label_s'], b['w_s'], b['x_t'], b['w_t'])          # warm-up
    log('cpu baseline warm-up done')
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.step(b['x_s'], b['label_s'], b['w_s'], b['x_t'], b['w_t'])
    dt = (time.perf_counter() - t0) / iters
    return {'value': round(2 * batch / dt, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': '%d full A+B+C iterations of %s at %dx%d, batch %d, fp32, %d threads (after 1 warm-up); '
                      '%.2f s/iteration' % (iters, arch, image_size, image_size, batch, cores, dt)}


def insitu_top(launches, iters, overhead_us, mfma_tflops, hbm_tbps=6.3, k=10):
    """Group the conv-family launches of the roofline pass by layer label; per label: launches / iteration, mean in-situ duration
    (event time minus dispatch latency), floor and gap.  Returns the k largest gaps and the table's total (ms / iteration)."""
    rows = {}
    for e in launches:
        if e['family'] != 0:
            continue
        r = rows.setdefault(e['label'], [0, 0.0, e['flops'], e['bytes']])
        r[0] += 1; r[1] += max(e['us'] - overhead_us, 0.0)
    out, total = [], 0.0
    for lab, (n, us, fl, by) in rows.items():
        mean = us / n
        floor = max(fl / (mfma_tflops * 1e12) * 1e6, by / (hbm_tbps * 1e12) * 1e6, 5.0)
        per_it = n / float(iters)
        total += mean * per_it
        out.append({'layer': lab, 'per_step': round(per_it, 2), 'us': round(mean, 1), 'floor_us': round(floor, 1),
                    'gap_ms_per_step': round((mean - floor) * per_it * 1e-3, 3)})
    out.sort(key=lambda r: -r['gap_ms_per_step'])
    return {'floor': 'max(FLOP / %.0f TFLOP/s, bytes / %.1f TB/s, 5 us)' % (mfma_tflops, hbm_tbps), 'layers': len(out),
            'table_ms_per_step': round(total * 1e-3, 3), 'top': out[:k]}


def main():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC for RCCL; must be set before HIP initialises
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    dev_index = local_rank if args.backend == 'nccl' else 0
    if world > 1 and args.backend != 'nccl':
        os.environ['MI355_BN_RESIDENT'] = '0'      # ranks share one GPU: no kernel may claim every CU for resident blocks
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')

    import mi355
    from mi355 import ops
    from mi355.da_step import build_training
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from utils.synthetic import make_batch

    mi355.load()
    mi355.set_compute_dtype(args.dtype)
    torch.manual_seed(1)                                   # reference default seed (train1.py:664)
    S, B = args.image_size, args.batch_size
    backbone = models.__dict__[args.arch](pretrained=False)   # random init: no checkpoints offline
    model = PoseResNetx9(backbone, Upsampling(backbone.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
    if world > 1:
        from mi355.da_step import broadcast_module
        broadcast_module(model, src=0)
    step, opts, scheds = build_training(model, heatmap_size=S // 4)
    # Random-noise images carry no pose: after ~40 iterations the target predictions of some image collapse onto one pixel and
    # the reference's per-map max-normalisation divides 0 by 0 (regda_7.py:3623-3625) -- NaN from there on, in the reference too.
    # The benchmark keeps such maps at zero so that long runs stay finite; same kernels, same work (flag off = reference rule).
    for c in step.crit.values():
        if hasattr(c, 'guard_empty_maps'):
            c.guard_empty_maps = True
    batch = make_batch(B, S, S // 4, seed=1 + rank, device=dev)

    def tick():
        for s in scheds.values():
            s.step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (eager), capture, warm-up (replay)
    n_eager = max(2, min(args.warmup, 3))
    for _ in range(n_eager):
        step.run(batch); tick()
    log('eager warm-up done')
    use_graph = (not args.no_graph) and (world == 1 or args.graph)
    if world > 1 and not use_graph and not args.no_graph and (args.backend == 'nccl' or os.environ.get('MI355_BENCH_ADAPT_ANY') == '1'):
        # eager launches keep the gradient exchange overlapped with the backward, but only pay off while the host can feed
        # the GPU: if enqueueing an iteration takes (nearly) as long as running it on ANY rank, every rank replays graphs
        # instead (collectives between the graphs).  The decision is all-reduced so that all ranks take the same path.
        use_graph = step.choose_launch_mode(batch, after=tick) == 'graph'
        log('multi-rank launch mode: host/GPU time ratio %.2f -> %s' % (step.host_gpu_ratio, 'graph replay' if use_graph else 'eager + overlapped all-reduce'))
    if use_graph:
        ok = 1
        try:
            step.capture(batch, warmup=0)
            if world > 1:                  # one trial replay before the timed region: a capture problem must not cost the run
                step.run(batch); tick(); torch.cuda.synchronize()
        except Exception as e:             # noqa: BLE001 -- any failure of the captured path falls back to eager launches
            ok = 0
            log('graph capture / trial replay failed (%s: %s): falling back to eager launches' % (type(e).__name__, e))
        if world > 1:
            f = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            ok = int(f)
        if ok:
            log('graphs captured' + (' (gradient exchange captured with them)' if getattr(step, 'exchange_captured', False) else ''))
        else:
            step.graphs, step.exchange_captured, use_graph = None, False, False
    for _ in range(max(0, args.warmup - n_eager)):
        step.run(batch); tick()

    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run(batch); tick()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    log('timed region done: %.2f ms/step' % ms_per_step)
    losses = {k: float(step.out[k]) for k in ('loss_s', 'loss_gf', 'loss_gt')}

    # ---- forward-only throughput of the test.py path (eval-mode BatchNorm, main head only, arg-max decode), SURVEY 8(d)
    eval_ms = None
    if not args.no_eval:
        model.eval()
        from utils.keypoint_detection import get_max_preds_device
        from mi355.infer import GraphedForward
        forward = GraphedForward(model)                     # what validate() of train1.py / test.py runs
        with torch.no_grad():
            for _ in range(4):
                get_max_preds_device(forward(batch['x_t']))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_eval = 20
            for _ in range(n_eval):
                get_max_preds_device(forward(batch['x_t']))
            torch.cuda.synchronize()
            eval_ms = (time.perf_counter() - t0) / n_eval * 1e3
        model.train()
        log('eval path: %.2f ms per batch of %d' % (eval_ms, B))

    # ---- roofline of the dominant kernel family (MFMA implicit-GEMM conv: gather + wgrad kernels), rank 0:
    # the same K steps again, eagerly, every conv launch bracketed by hipEvents on its stream.
    roof = None
    if not args.no_roofline:
        # every rank runs these steps (they contain the gradient all-reduces); rank 0's events are the ones reported
        step.graphs = None
        n_prof = min(args.steps, 5)
        ops.prof_reset(); ops.prof_enable(True)
        # eager launches are host-bound here (the Python layer cannot feed a 40 ms iteration in 40 ms), and an event pair
        # around a launch the GPU had to wait for would also time the wait: so every iteration is enqueued behind an idle
        # spin as long as the host needs to enqueue it -- the GPU then runs launch after launch, events back to back
        torch.cuda.synchronize()
        t_q = time.perf_counter(); step.run(batch); tick(); host_ms = (time.perf_counter() - t_q) * 1e3
        torch.cuda.synchronize()
        ops.prof_reset()
        for _ in range(n_prof):
            ops.spin_us(int(min(1.5 * host_ms + 20.0, 1500.0) * 1e3))
            step.run(batch); tick()
            torch.cuda.synchronize()
        ops.prof_enable(False)
        log('roofline pass: host needs %.0f ms to enqueue one eager iteration' % host_ms)
    if not args.no_roofline and rank == 0:
        ms, launches, flops, abytes = ops.prof_read()
        log('roofline pass done')
        # executed -> algorithmic FLOPs of the 3-channel stem (K = 147): it runs as a 4x4 conv over the 2x2 space-to-depth image
        # (K = 16 taps x 16 folded channels = 256) or, with odd extents / MI355_STEM_S2D=0, as the 7x7 conv over the image padded to
        # one 16-byte chunk (K = 49 x 8 bf16 / 49 x 4 fp32 channels)
        cpad = 4 if args.dtype == 'f32' else 8
        stem_m = B * (S // 2) * (S // 2)
        stem_k = 256 if (getattr(model.backbone.conv1, '_s2d', False) and S % 2 == 0) else 49 * cpad
        stem_excess = 2.0 * stem_m * 64 * (stem_k - 147) * 4 * n_prof      # 2 fwd (A, shared B+C) + 2 wgrad (A, C) launches / iteration
        algo = flops - stem_excess
        ach = algo / (ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        roof = {'bound': 'mfma', 'kernel': 'gather_gemm_kernel%s + wgrad_gemm_kernel + wgrad_kw_kernel (implicit-GEMM conv family)' % (' + gather_fp8_kernel' if args.dtype == 'fp8' else ''),
                'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': None, 'traffic_unit': 'bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes)',
                'algorithmic_bytes_per_launch': round(abytes / launches), 'launches_per_step': launches // n_prof,
                'algorithmic_gflop_per_launch': round(algo / launches / 1e9, 3),
                'avg_launch_us': round(ms * 1e3 / launches, 2),
                'event_dispatch_overhead_us': round(ops.prof_event_overhead_us(256), 2),
                'conv_ms_per_step': round(ms / n_prof, 3)}
        # the family mixes regimes: launches whose algorithmic intensity is below the machine balance (2.5 PFLOP/s / 8 TB/s =
        # 312 FLOP/B: the 1x1 convs, the stem) are HBM-bound and priced against 8 TB/s, the others against MFMA
        sp = ops.prof_read_split(PEAK_TFLOPS[args.dtype] * 1e12 / 8e12)
        if sp[0] > 0 and sp[4] > 0:
            roof['split'] = {
                'hbm_bound': {'launches_per_step': int(sp[3]) // n_prof, 'ms_per_step': round(sp[0] / n_prof, 3),
                              'achieved_GBps_algorithmic': round(sp[2] / (sp[0] * 1e-3) / 1e9, 1), 'peak_GBps': 8000.0,
                              'frac': round(sp[2] / (sp[0] * 1e-3) / 8e12, 4)},
                'mfma_bound': {'launches_per_step': int(sp[7]) // n_prof, 'ms_per_step': round(sp[4] / n_prof, 3),
                               'achieved_TFLOPs': round(sp[5] / (sp[4] * 1e-3) / 1e12, 1), 'peak_TFLOPs': peak,
                               'frac': round(sp[5] / (sp[4] * 1e-3) / 1e12 / peak, 4)}}
        # in-situ layer table of the family: the launches of the pass above grouped by layer label, each against its own floor
        # max(FLOP / 1.75 PFLOP/s [the dense bf16 rate at the 1.67 GHz the chip holds under MFMA load], bytes / 6.3 TB/s [achievable
        # HBM], 5 us); event times minus the calibrated dispatch latency; the ten largest gaps (ms per iteration above the floor)
        roof['insitu_top'] = insitu_top(ops.prof_launches(), n_prof, roof['event_dispatch_overhead_us'], PEAK_TFLOPS[args.dtype] * 0.7)
        if args.launch_log and world == 1:
            ops.prof_reset(); ops.prof_enable(2)
            ops.spin_us(int(min(1.5 * host_ms + 20.0, 1500.0) * 1e3))
            step.run(batch); tick()
            torch.cuda.synchronize()
            ops.prof_enable(False)
            json.dump({'event_dispatch_overhead_us': roof['event_dispatch_overhead_us'], 'launches': ops.prof_launches()}, open(args.launch_log, 'w'))
    if world > 1:
        dist.barrier()

    if rank == 0:
        F = F_GFLOP.get((args.arch, S))
        res = {
            'metric': 'images/sec (unique source+target images per A+B+C training iteration, 2*B*N/t_iter)',
            'value': round(2 * B * world / (ms_per_step * 1e-3), 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'data_note': 'all-zero ground-false maps stay zero instead of the reference 0/0 = NaN (DESIGN.md section 7)',
            'config': {'workload': 'RHD->H3D domain-adaptation iteration (steps A+B+C), %s + 3-deconv neck + main head + 3 '
                                   'adversarial multiscale-fusion heads, %dx%d, batch %d source + %d target per GPU, '
                                   'random init, synthetic batches' % (args.arch, S, S, B, B),
                       'arch': args.arch, 'image_size': S, 'per_gpu_batch': B, 'global_batch': B * world,
                       'parallelism': 'dp%d' % world, 'hip_graphs': use_graph,
                       'gradient_exchange': ('none (one rank)' if world == 1 else 'RCCL all-reduce captured into the graphs, overlapped with the backward'
                                             if getattr(step, 'exchange_captured', False) else 'between the graphs (blocking)' if use_graph
                                             else 'eager, overlapped with the backward'),
                       **({'fp8_gemms': 'forward + input gradient of the 3x3 convs'
                           + (', transposed convs' if os.environ.get('MI355_FP8_DECONV', '0') == '1' else '')
                           + (', weight gradients' if os.environ.get('MI355_FP8_WGRAD', '0') == '1' else '')}
                          if args.dtype == 'fp8' else {})},
            'model_passes_per_s': round(3 * B * world / (ms_per_step * 1e-3), 2),
            'eval_images_per_s_per_gpu': round(B / (eval_ms * 1e-3), 1) if eval_ms else None,
            'losses_last_step': losses,
            'bn_resident_timeouts': ops.bn_resident_timeouts(),      # 0 = every one-launch BatchNorm backward had all its blocks resident
        }
        if F is not None:
            tf = 9 * B * F * 1e9 / (ms_per_step * 1e-3) / 1e12
            res['iteration_tflops_per_gpu_algorithmic_9BF'] = round(tf, 2)
            res['iteration_frac_of_mfma_peak'] = round(tf / PEAK_TFLOPS[args.dtype], 4)
        if roof is not None:
            cfg, src = pmc_record(args)
            if cfg is not None:
                fam = cfg['families']
                roof['traffic'] = cfg.get('conv_family_bytes_per_launch')
                roof['traffic_source'] = src
                # counter-based HBM rates of the memory-bound stages (north star: "achieved HBM GB/s on the BN / soft-argmax
                # stages"): measured bytes / measured kernel time per family, with the algorithmic bytes beside them
                roof['membound'] = {k: {kk: v[kk] for kk in ('launches', 'bytes_per_launch', 'counter_GBps', 'frac_of_8TBps',
                                                             'algorithmic_bytes_per_launch', 'traffic_over_algorithmic') if kk in v}
                                    for k, v in fam.items() if k in ('bn_fwd', 'bn_bwd', 'kl_loss', 'argmax', 'softargmax',
                                                                     'pseudo_label', 'slab_reduce', 'pointwise21', 'optimizer')}
                if 'conv_mfma' in fam and 'traffic_over_algorithmic' in fam['conv_mfma']:
                    roof['traffic_over_algorithmic'] = fam['conv_mfma']['traffic_over_algorithmic']
            res['roofline'] = roof
        if not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline(args.arch, S, args.cpu_baseline_batch, args.cpu_baseline_iters)
            except IndexError:
                # the reference's pseudo-label tables are fixed to 64x64 heat-maps (regda_7.py:2956-3039, 3118-3201): its
                # restatement cannot run other input sizes; the CPU figure is then taken at the reference's own 256x256
                res['cpu_baseline'] = cpu_baseline(args.arch, 256, args.cpu_baseline_batch, args.cpu_baseline_iters)
                res['cpu_baseline']['sample'] += ' -- at 256x256: the reference fixes its pseudo-label tables to 64x64 heat-maps'
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
