#!/usr/bin/env python
"""microseconds of the fc8-SEC heads kernels at the train-s shape (B=16, 41x41, 4 x 1024 -> 21)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
B, K, H, W, O, n = 16, 1024, 41, 41, 21, 4
xs = [torch.randn(B, K, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(n)]
w = torch.randn(n, O, K, device="cuda") * 0.05
b = torch.randn(n, O, device="cuda")
g = torch.randn(B, O, H, W, device="cuda")
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print("heads_forward  %.1f us" % t(lambda: ops.heads_forward(xs, w, b)))
print("heads_backward %.1f us (dx + dw + reduce)" % t(lambda: ops.heads_backward(xs, w, g)))
print("heads_backward %.1f us (dw only)" % t(lambda: ops.heads_backward(xs, w, g, need_gx=False)))
