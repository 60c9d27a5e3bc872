#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes of the direct convolution kernels (profiles/pmc_traffic_conv.json): the byte-count
calibration launch of pmc_workload.py (softmax over a tensor larger than the Infinity Cache), then every direct kernel three
times at the train-s shapes (batch 16): forward / data-gradient kernels, weight-gradient kernels and their reductions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops  # noqa: E402

cl = torch.channels_last
big = torch.randn(2048, 21, 41, 41, device="cuda")          # 289 MB
for _ in range(3):
    ops.softmax_forward(big)                                  # calibration: softmax_fwd_kernel
torch.cuda.synchronize()
del big
for cin, cout, hw in [(3, 64, 321), (64, 64, 321), (64, 128, 161), (128, 128, 161), (128, 64, 161)]:
    x = torch.randn(16, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=cl)
    b = torch.randn(cout, device="cuda")
    g = torch.randn(16, cout, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    for _ in range(3):
        ops.conv3x3_direct(x, w, b, True)
        if (cin, cout) in ops.WGRAD_CONV_SHAPES:
            ops.conv3x3_wgrad(x, g)
    torch.cuda.synchronize()
    print("%d -> %d @ %d: in %d B, out %d B" % (cin, cout, hw, x.numel() * 2, g.numel() * 2))
