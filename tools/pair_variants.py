import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "tf-nas_amd"))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search
B = int(sys.argv[1]); search.INTERLEAVE_PATHS = sys.argv[2] == '1'; search.HOST_SAMPLING = sys.argv[3] == '1'
dev = torch.device("cuda", 0)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup("gpu")).to(dev); model.set_temperature(5.0)
state = search.SearchState(model); opt_w, opt_a = search.make_optimizers(model); noise = search.NoiseSource(2)
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 100, (B,), device=dev)
for i in range(4): search.search_iteration_pair(state, opt_w, opt_a, ((x, y), (x, y)), (x, y), noise)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 12
for i in range(n): search.search_iteration_pair(state, opt_w, opt_a, ((x, y), (x, y)), (x, y), noise)
torch.cuda.synchronize()
print("B=%d interleave=%s host_sampling=%s  %.2f ms/pair" % (B, sys.argv[2], sys.argv[3], (time.perf_counter() - t0) / n * 1e3))
