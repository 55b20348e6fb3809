import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch, torch.distributed as dist
from tfnas_amd import Network, load_lat_lookup, geometry, search
torch.cuda.set_device(0); dev = torch.device('cuda', 0)
import os as _os
print('affinity', len(_os.sched_getaffinity(0)), 'OMP', _os.environ.get('OMP_NUM_THREADS'), 'torch threads', torch.get_num_threads())
if _os.environ.get('NO_PG') != '1':
    dist.init_process_group('nccl', device_id=dev)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev); model.set_temperature(5.0)
state = search.SearchState(model); opt_w, opt_a = search.make_optimizers(model); noise = search.NoiseSource(2)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 100, (128,), device=dev)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for force in ((False, False) if _os.environ.get('NO_PG') == '1' else (False, True)):
    search.FORCE_ALLREDUCE_AT_WORLD_1 = force
    w = t(lambda: search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos()))
    a = t(lambda: search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev)))
    print('force_allreduce=%s  w_step %.2f ms  a_step %.2f ms  OMP=%s' % (force, w, a, os.environ.get('OMP_NUM_THREADS')))
if dist.is_initialized(): dist.destroy_process_group()
