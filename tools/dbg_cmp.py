import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
w = 0
for k in a['after_w']:
    x, y = a['after_w'][k].double(), b['after_w'][k].double()
    if x.numel() and x.dtype.is_floating_point:
        e = float((x - y).abs().max()); r = float(x.abs().max())
        if e > 1e-5 + 5e-5 * r: print('after_w', k, e, r)
for i, (x, y) in enumerate(zip(a['grads'], b['grads'])):
    print('grad', i, float((x - y).abs().max()), float(x.abs().max()))
