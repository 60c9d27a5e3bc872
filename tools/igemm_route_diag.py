#!/usr/bin/env python
"""per-parameter gradient agreement of the implicit-GEMM backbone route, the im2col route and a float32 backbone (same weights,
input and output gradient, dropout off): relative L2 distances, layer by layer"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsrg_amd import backbone

CL = torch.channels_last
torch.manual_seed(3)
net = backbone.VGG16ASPP(dropout=0.0).cuda().to(memory_format=CL)
x = torch.randn(2, 3, 161, 161, device="cuda").contiguous(memory_format=CL)
res, gout = {}, None
for tag, route, amp in (("igemm", True, torch.bfloat16), ("im2col", False, torch.bfloat16), ("fp32", False, None)):
    backbone._IGEMM = route
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=amp or torch.bfloat16, enabled=amp is not None):
        y = net(x)
    if gout is None:
        gout = torch.randn_like(y)
    y.backward(gout)
    res[tag] = (y.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in net.named_parameters()})
backbone._IGEMM = True
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-20))
print("scores: igemm-vs-im2col %.3e  igemm-vs-fp32 %.3e  im2col-vs-fp32 %.3e" % (
    rel(res["igemm"][0], res["im2col"][0]), rel(res["igemm"][0], res["fp32"][0]), rel(res["im2col"][0], res["fp32"][0])))
for n in res["fp32"][1]:
    a, b, c = res["igemm"][1][n], res["im2col"][1][n], res["fp32"][1][n]
    print("%-28s igemm-vs-im2col %.3e  igemm-vs-fp32 %.3e  im2col-vs-fp32 %.3e  |g| %.3e" % (n, rel(a, b), rel(a, c), rel(b, c), float(c.norm())))
