#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes over the implicit-GEMM convolution kernels: conv4_2's forward (512 -> 512 at 41x41, batch
16), the four fc6_k in one launch, conv3_2 at 81x81, and their weight gradients, five launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops  # noqa: E402

cl = torch.channels_last
for name, H, cin, cout, dils in [("conv4_2", 41, 512, 512, [1]), ("fc6x4", 41, 512, 1024, [6, 12, 18, 24]), ("conv3_2", 81, 256, 256, [1])]:
    n = len(dils)
    xs = [torch.randn(16, cin, H, H, device="cuda").bfloat16().contiguous(memory_format=cl) for _ in range(n)]
    ws = [ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)) for _ in range(n)]
    bs = [torch.randn(cout, device="cuda") for _ in range(n)]
    gs = [torch.randn(16, cout, H, H, device="cuda").bfloat16().contiguous(memory_format=cl) for _ in range(n)]
    for _ in range(5):
        ops.conv_igemm(xs, ws, bs, dils, 3, True)
    for _ in range(5):
        ops.conv_igemm_wgrad(xs, gs, dils, 3)
    torch.cuda.synchronize()
    print(name, "done")
