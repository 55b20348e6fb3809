import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search
dev = torch.device('cuda', 0)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev); model.set_temperature(5.0)
state = search.SearchState(model); opt_w, opt_a = search.make_optimizers(model); noise = search.NoiseSource(2)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 100, (128,), device=dev)
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k0 = state._alpha_key()
    a = state.alpha_host(); t1 = time.perf_counter()
    e = noise.exp(dev); t2 = time.perf_counter()
    pos = search.host_gumbel_positions(a, e._tfnas_host, 5.0); t3 = time.perf_counter()
    search.w_step(state, x, y, opt_w, 5.0, e, noise.rand_pos()); t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    ms_ = torch.cuda.memory_stats(); print('dev_allocs %d reserved %.1f GB  ' % (ms_['num_device_alloc'], ms_['reserved_bytes.all.current'] / 2**30), end='')
    print('alpha_host %.2f ms  exp %.2f  pos %.2f  w_step enqueue %.2f  total %.2f  key_changed %s threads %d' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t3)*1e3, k0 != state._alpha_key(), torch.get_num_threads()))
