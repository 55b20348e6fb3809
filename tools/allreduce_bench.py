"""Host/GPU cost of search.allreduce_mean_ on the w-step's gradient list (run under torchrun, 1+ ranks)."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch, torch.distributed as dist
from tfnas_amd import Network, load_lat_lookup, geometry, search
rank = int(os.environ.get('RANK', 0)); lr = int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(lr)
dist.init_process_group('nccl')
search.FORCE_ALLREDUCE_AT_WORLD_1 = True
dev = torch.device('cuda', lr)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
ws = model.weight_parameters()
# a sampled path touches 1 of 8 candidates per cell: take every 8th MBConv-ish tensor + stems/head
grads = [torch.randn_like(p) for i, p in enumerate(ws) if i % 8 == 0]
print('tensors', len(grads), 'MB', sum(g.numel() for g in grads) * 4 / 1e6)
for _ in range(3): search.allreduce_mean_(grads)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): search.allreduce_mean_(grads)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host enqueue %.2f ms / call, incl. GPU drain %.2f ms / call' % ((t1 - t0) * 100, (t2 - t0) * 100))
dist.destroy_process_group()
