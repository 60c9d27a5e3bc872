#!/usr/bin/env python
"""microseconds of the narrow 3x3 layers at batch 16 (conv1_2 64 -> 64 at 321x321; conv2_1 64 -> 128, conv2_2 128 -> 128 and
the data gradient of conv2_1, 128 -> 64, at 161x161): the direct HIP kernel against MIOpen through F.conv2d"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
cl = torch.channels_last
def t(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for cin, cout, hw in [(64, 64, 321), (64, 128, 161), (128, 128, 161), (128, 64, 161)]:
    x = torch.randn(16, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=cl)
    b = torch.randn(cout, device="cuda")
    flops = 2 * 16 * hw * hw * cout * 9 * cin
    byts = 16 * hw * hw * (cin + cout) * 2
    us = t(lambda: ops.conv3x3_direct(x, w, b, True))
    print("%3d -> %3d @ %d: direct kernel  %.1f us  %.0f TFLOP/s  %.2f TB/s of in+out" % (cin, cout, hw, us, flops / us / 1e6, byts / us / 1e6))
    us = t(lambda: torch.relu_(F.conv2d(x, w, b.bfloat16(), padding=1)))
    print("%3d -> %3d @ %d: F.conv2d+relu  %.1f us  %.0f TFLOP/s" % (cin, cout, hw, us, flops / us / 1e6))
print("weight gradients")
for cin, cout, hw in [(64, 64, 321), (64, 128, 161), (128, 128, 161)]:
    x = torch.randn(16, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    g = torch.randn(16, cout, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=cl)
    flops = 2 * 16 * hw * hw * cout * 9 * cin
    us = t(lambda: ops.conv3x3_wgrad(x, g))
    print("%3d -> %3d @ %d: direct wgrad  %.1f us  %.0f TFLOP/s  %.2f TB/s of x+g" % (cin, cout, hw, us, flops / us / 1e6, 16 * hw * hw * (cin + cout) * 2 / us / 1e6))
    us = t(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    print("%3d -> %3d @ %d: MIOpen wrw    %.1f us  %.0f TFLOP/s" % (cin, cout, hw, us, flops / us / 1e6))
x = torch.randn(16, 3, 321, 321, device="cuda").bfloat16().contiguous(memory_format=cl)
w = (torch.randn(64, 3, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=cl)
b = torch.randn(64, device="cuda")
g = torch.randn(16, 64, 321, 321, device="cuda").bfloat16().contiguous(memory_format=cl)
print("  3 ->  64 @ 321: direct kernel %.1f us, F.conv2d + bias + relu %.1f us; direct wgrad %.1f us, MIOpen wrw %.1f us" % (
    t(lambda: ops.conv3x3_direct(x, w, b, True)), t(lambda: torch.relu_(F.conv2d(x, w, b.bfloat16(), padding=1))),
    t(lambda: ops.conv3x3_wgrad(x, g)),
    t(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))))
