#!/usr/bin/env python
"""One-off audit of the PyTorch-ROCm ops the plumbing relies on, channels_last bf16 on the GPU against fp32 on the CPU
(found: avg_pool2d backward is wrong for channels_last inputs in PyTorch 2.10+rocm7.0 — replaced by our own kernel)."""
import torch, torch.nn.functional as F
cl = torch.channels_last
torch.manual_seed(0)

def rel(a, b):
    return ((a.float().cpu() - b).norm() / (b.norm() + 1e-12)).item()

def check(name, fn, x, *params):
    xs = x.detach().clone().requires_grad_(True)
    ps = [p.detach().clone().requires_grad_(True) for p in params]
    y = fn(xs, *ps)
    g = torch.randn_like(y)
    y.backward(g)
    xg = x.detach().cuda().bfloat16().contiguous(memory_format=cl).requires_grad_(True)
    pg = [p.detach().cuda().bfloat16().requires_grad_(True) for p in params]
    yg = fn(xg, *pg)
    yg.backward(g.cuda().bfloat16().contiguous(memory_format=cl))
    msg = "%-34s fwd %.4f  dx %.4f" % (name, rel(yg, y.detach()), rel(xg.grad, xs.grad))
    for i, (a, b) in enumerate(zip(pg, ps)):
        msg += "  dp%d %.4f" % (i, rel(a.grad, b.grad))
    print(msg, flush=True)

x = torch.randn(2, 64, 41, 41)
w = torch.randn(96, 64, 3, 3) * 0.05
b = torch.randn(96) * 0.1
for d in (1, 2, 12):
    check("conv2d 3x3 dilation %d" % d, lambda x, w, b, d=d: F.conv2d(x, w, b, 1, d, d), x, w, b)
check("conv2d 1x1", lambda x, w, b: F.conv2d(x, w, b), x, torch.randn(32, 64, 1, 1) * 0.1, torch.randn(32) * 0.1)
check("conv2d 7x7 stride 2", lambda x, w: F.conv2d(x, w, None, 2, 3), torch.randn(2, 8, 65, 65), torch.randn(16, 8, 7, 7) * 0.05)
check("max_pool2d 3/2/1 ceil", lambda x: F.max_pool2d(x, 3, 2, 1, ceil_mode=True), x)
check("max_pool2d 3/1/1", lambda x: F.max_pool2d(x, 3, 1, 1), x)
check("avg_pool2d 3/1/1", lambda x: F.avg_pool2d(x, 3, 1, 1), x)
check("relu", lambda x: F.relu(x), x)
check("interpolate bilinear align", lambda x: F.interpolate(x, size=(81, 77), mode="bilinear", align_corners=True), x)
check("add + mul", lambda x: x * 0.5 + x, x)
check("log_softmax dim 1", lambda x: F.log_softmax(x, 1), x)
