#!/usr/bin/env python
"""the direct convolution's data gradient with the ReLU backward of the layer below in its store against the plain launch + ops.relu_bwd_bias
(conv1_2 -> conv1_1 at 321x321, conv2_2 -> conv2_1 at 161x161, batch 16):  python tools/direct_dgrad_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops                                                           # noqa: E402

CL = torch.channels_last


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("%-24s %9s %9s %9s" % ("layer", "fused", "separate", "dgrad only"))
for name, c, H in (("conv1_2 -> conv1_1 out", 64, 321), ("conv2_2 -> conv2_1 out", 128, 161)):
    g = torch.randn(16, c, H, H, device="cuda").bfloat16().contiguous(memory_format=CL)
    y = torch.relu(torch.randn(16, c, H, H, device="cuda")).bfloat16().contiguous(memory_format=CL)
    wt = (torch.randn(c, c, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=CL)
    fns = {"f": lambda: ops.conv3x3_direct_dgrad(g, wt, y), "s": lambda: ops.relu_bwd_bias(ops.conv3x3_direct(g, wt, None, False), y, 1.0),
           "p": lambda: ops.conv3x3_direct(g, wt, None, False)}
    for fn in fns.values():
        for _ in range(3):
            fn()
    t = {k: [] for k in fns}
    for _ in range(5):
        for k, fn in fns.items():
            t[k].append(timed(fn))
    print("%-24s %9.1f %9.1f %9.1f" % (name, np.median(t["f"]), np.median(t["s"]), np.median(t["p"])), flush=True)
