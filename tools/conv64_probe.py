#!/usr/bin/env python
"""microseconds of conv1_2 (64 -> 64, 3x3, 321x321, batch 16): the direct HIP kernel against MIOpen through F.conv2d"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
cl = torch.channels_last
x = torch.randn(16, 64, 321, 321, device="cuda").bfloat16().contiguous(memory_format=cl)
w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=cl)
b = torch.randn(64, device="cuda")
def t(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
flops = 2 * 16 * 321 * 321 * 64 * 576
us = t(lambda: ops.conv3x3_c64(x, w, b, True))
print("direct kernel  %.1f us  %.0f TFLOP/s  %.2f TB/s of in+out" % (us, flops / us / 1e6, 2 * x.numel() * 2 / us / 1e6))
for flags, what in [(3, "no global stores"), (5, "no prefetch/park"), (9, "no MFMAs"), (15, "sync + epilogue only")]:
    print("  [%s] %.1f us" % (what, t(lambda: ops.conv3x3_c64(x, w, b, flags))))
us = t(lambda: torch.relu_(F.conv2d(x, w, b.bfloat16(), padding=1)))
print("F.conv2d+relu  %.1f us  %.0f TFLOP/s" % (us, flops / us / 1e6))
