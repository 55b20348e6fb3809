import os, sys, cProfile, pstats
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
model.set_temperature(5.0)
state = search.SearchState(model)
opt_w, opt_a = search.make_optimizers(model)
noise = search.NoiseSource(2)
x = torch.randn(128, 3, 224, 224, device=dev); y = torch.randint(0, 100, (128,), device=dev)
def pair():
    search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos())
    search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev))
for _ in range(3): pair()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): pair()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(45)
