#!/usr/bin/env python3
"""Micro-benchmark of stem/head formulations with stock PyTorch-ROCm ops (GPU box)."""
import time, torch, torch.nn.functional as F
dev = 'cuda'
B = 128
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
x = torch.randn(B, 3, 224, 224, device=dev)
w1 = torch.randn(32, 3, 3, 3, device=dev, requires_grad=True)
def stem1_nchw():
    y = F.conv2d(x, w1, None, 2, 1); y.sum().backward()
print('first_stem conv NCHW fwd+wgrad %.2f ms' % t(stem1_nchw))
xcl = x.contiguous(memory_format=torch.channels_last)
def stem1_unfold():
    cols = F.unfold(x, 3, padding=1, stride=2)            # [B, 27, 12544]
    y = torch.matmul(w1.view(32, 27), cols)               # [B, 32, 12544]
    y.sum().backward()
print('first_stem unfold+matmul fwd+wgrad %.2f ms' % t(stem1_unfold))
h = torch.randn(B, 32, 112, 112, device=dev, requires_grad=True)
wd = torch.randn(32, 1, 3, 3, device=dev, requires_grad=True)
def dw_nchw():
    y = F.conv2d(h, wd, None, 1, 1, 1, 32); y.sum().backward()
print('second_stem depthwise NCHW fwd+dgrad+wgrad %.2f ms' % t(dw_nchw))
hcl = h.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
def dw_shift():
    # depthwise 3x3 as 9 shifted multiply-adds (pure elementwise; autograd gives wgrad as reductions)
    p = F.pad(hcl, (1, 1, 1, 1))
    y = 0
    for ky in range(3):
        for kx in range(3):
            y = y + p[:, :, ky:ky + 112, kx:kx + 112] * wd[:, 0, ky, kx].view(1, 32, 1, 1)
    y.sum().backward()
print('second_stem depthwise as 9 shifted FMAs (channels_last) %.2f ms' % t(dw_shift))
wp = torch.randn(16, 32, 1, 1, device=dev, requires_grad=True)
def pw_nchw():
    y = F.conv2d(h, wp); y.sum().backward()
print('second_stem 1x1 project NCHW fwd+bwd %.2f ms' % t(pw_nchw))
def pw_mm():
    y = hcl.permute(0, 2, 3, 1).reshape(-1, 32) @ wp.view(16, 32).t(); y.sum().backward()
print('second_stem 1x1 project as matmul on NHWC fwd+bwd %.2f ms' % t(pw_mm))
def bn_nchw():
    y = F.batch_norm(h, None, None, None, None, True, 0.0, 1e-5); y.sum().backward()
print('BN NCHW 32ch 112^2 fwd+bwd %.2f ms' % t(bn_nchw))
def bn_cl():
    y = F.batch_norm(hcl, None, None, None, None, True, 0.0, 1e-5); y.sum().backward()
print('BN channels_last fwd+bwd %.2f ms' % t(bn_cl))
f = torch.randn(B, 320, 7, 7, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
wf = torch.randn(1280, 320, 1, 1, device=dev, requires_grad=True)
def head_conv():
    y = F.conv2d(f, wf); y.sum().backward()
print('feature_mix 1x1 conv channels_last fwd+bwd %.2f ms' % t(head_conv))
def head_mm():
    y = f.permute(0, 2, 3, 1).reshape(-1, 320) @ wf.view(1280, 320).t(); y.sum().backward()
print('feature_mix as matmul fwd+bwd %.2f ms' % t(head_mm))
