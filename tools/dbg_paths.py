#!/usr/bin/env python3
"""Debug aid: compare the path level with the per-cell route piece by piece (forward outputs, every gradient)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
import torch.nn.functional as F
from tfnas_amd import Network, load_lat_lookup, geometry, search

lut = load_lat_lookup('gpu')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def model():
    torch.manual_seed(2)
    m = Network(100, geometry.initial_mc_num_dddict(), lut).cuda()
    m.set_temperature(5.0)
    return m


def diff(tag, a, b):
    if a is None or b is None:
        print('%-60s %s' % (tag, 'None mismatch' if (a is None) != (b is None) else 'both None'))
        return
    d = float((a - b).abs().max())
    print('%-60s max|d| %.3e  (ref max %.3e)%s' % (tag, d, float(b.abs().max()), '' if d == 0 else '   <<<'))


g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(B, 3, 224, 224, device='cuda', generator=g)
y = torch.randint(0, 100, (B,), device='cuda', generator=g)
idx_a = [i % 8 for i in range(18)]
idx_b = [(i + 3) % 8 for i in range(18)]

# ---------------- sampled single path, forward + backward
ma, mb = model(), model()
sa = search.SearchState(ma)
sa.require(True, False)
for p in mb.arch_parameters():
    p.requires_grad = False
sa.begin_weight_grads()
fa = ma._stem(x)
oa = sa.runner.sampled(fa, idx_a)
la = ma.classifier(ma._head(oa))
F.cross_entropy(la, y).backward()
sa.expose_weight_grads([idx_a])
# per-cell route with the same candidates
fb = mb._stem(x)
h = fb
ci = 0
from tfnas_amd.functions import MixedOpFn, SinkFn
for st in mb.stages():
    rs = [h]
    for blk in st.blocks():
        rs.append(MixedOpFn.apply(blk._plan((idx_a[ci],)), rs[-1], None, *blk.m_ops[idx_a[ci]].hip_params()))
        ci += 1
    h, _ = SinkFn.apply(st.betas, None, *rs[st.start_res:])
diff('single path: path output', oa, h)
lb = mb.classifier(mb._head(h))
diff('single path: logits', la, lb)
F.cross_entropy(lb, y).backward()
torch.cuda.synchronize()
bad = 0
for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
    if (pa.grad is None) != (pb.grad is None):
        print('grad presence differs', k, pa.grad is None, pb.grad is None)
        bad += 1
    elif pa.grad is not None and not torch.equal(pa.grad, pb.grad):
        if bad < 25:
            diff('grad ' + k, pa.grad, pb.grad)
        bad += 1
print('single sampled path: %d parameters with different gradients' % bad)

# ---------------- soft path
ma, mb = model(), model()
sa = search.SearchState(ma)
sa.require(False, True)
for p in mb.weight_parameters():
    p.requires_grad = False
e = torch.empty(18, 8, device='cuda').exponential_(generator=g)
la, lata = search._a_forward_paths(sa, x, e)
lb, latb = mb(x, False, exp_noise=e)
diff('soft: logits', la, lb)
diff('soft: lat', lata, latb)
(F.cross_entropy(la, y) + torch.abs(lata / 15 - 1) * 0.1).backward()
(F.cross_entropy(lb, y) + torch.abs(latb / 15 - 1) * 0.1).backward()
torch.cuda.synchronize()
for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
    if pa.grad is not None or pb.grad is not None:
        if pa.grad is None or pb.grad is None or not torch.equal(pa.grad, pb.grad):
            diff('soft grad ' + k, pa.grad, pb.grad)
print('soft done')
