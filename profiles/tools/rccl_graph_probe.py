import os, torch, torch.distributed as dist
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0); dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 20, device=dev); u = torch.ones(1 << 20, device=dev)
dist.all_reduce(t); torch.cuda.synchronize()      # warm-up: communicator creation outside capture
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g):
        t.mul_(2.0)
        w = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)     # on NCCL's stream, beside what follows
        u.add_(1.0)                                                      # 'rest of the backward'
        w.wait()
        t.add_(u)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('captured collective ok:', float(t[0]), float(u[0]))
dist.destroy_process_group()
