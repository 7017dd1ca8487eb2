"""Helper of tests/test_gpu_ddp.py::test_rccl_single_rank_smoke: six training iterations of a small model, with (argv[1] == '1') or
without a one-rank RCCL process group whose collectives are forced on (MI355_DDP_FORCE_COLLECTIVES, set by the test).  Prints one
JSON line: per-parameter checksums, the number of overlapped collective launches, the launch mode chosen."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, 'domain-adaptative-hand-pose-estimation_amd')]


def main():
    use_dist = sys.argv[1] == '1'
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    if use_dist:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)       # nccl IS RCCL on ROCm
    import mi355
    import mi355.da_step as ds
    import uda.model as models
    from uda.model.pose_resnet2 import Upsampling
    from uda.model.regda_7 import PoseResNetx9
    from utils.synthetic import make_batch
    mi355.load()
    mi355.set_compute_dtype('bf16')
    torch.manual_seed(1)
    bb = models.resnet18(pretrained=False)
    model = PoseResNetx9(bb, Upsampling(bb.out_features), 256, 21, num_head_layers=2, finetune=True).to(dev)
    ds.broadcast_module(model, 0)
    step, opts, scheds = ds.build_training(model, heatmap_size=32)
    batch = make_batch(2, 128, 32, seed=1, device=dev)
    launched = []
    orig = ds._OverlapReducer._launch
    ds._OverlapReducer._launch = lambda self, G, lo, hi: (launched.append((lo, hi)), orig(self, G, lo, hi))[1]
    tick = lambda: [s.step() for s in scheds.values()]
    for _ in range(3):                      # eager: gradient ranges all-reduced (AVG, async) from the backward's hooks
        step.run(batch)
        tick()
    mode = step.choose_launch_mode(batch, after=tick)      # includes the MAX all-reduce of the decision
    n_eager = len(launched)
    # graph replay: with RCCL the overlapped exchange is captured into the graphs (MI355_DDP_GRAPH_OVERLAP=0: blocking all-reduces
    # between the six graphs instead)
    step.capture(batch, warmup=0)
    n_captured = len(launched) - n_eager
    for _ in range(2):
        step.run(batch)
        tick()
    torch.cuda.synchronize()
    cs = [float(p.detach().double().abs().sum()) for p in model.parameters()]
    backend = dist.get_backend() if use_dist else None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    print(json.dumps({'checksums': cs, 'collective_launches': len(launched), 'captured_launches': n_captured, 'mode': mode, 'backend': backend,
                      'exchange_captured': bool(getattr(step, 'exchange_captured', False))}))


if __name__ == '__main__':
    main()
