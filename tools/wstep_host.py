#!/usr/bin/env python3
"""Host enqueue time vs GPU time of the search steps (path level vs per-cell route): TFNAS_PATHS=0/1 tools/wstep_host.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
if os.environ.get('WITH_RCCL') == '1':            # diagnostics: does an initialised process group change the step?
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29581')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, **({'device_id': dev} if os.environ.get('EAGER', '1') == '1' else {}))
    if os.environ.get('WARM_COLL') == '1':
        dist.all_reduce(torch.ones(4, device=dev)); torch.cuda.synchronize()
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
model.set_temperature(5.0)
state = search.SearchState(model)
opt_w, opt_a = search.make_optimizers(model)
noise = search.NoiseSource(2)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 100, (B,), device=dev)
for it in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos())
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev))
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 4:
        print('w_step host %.2f ms total %.2f ms | a_step host %.2f ms total %.2f ms' % (
            (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3), flush=True)
