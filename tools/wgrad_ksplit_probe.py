#!/usr/bin/env python
"""weight-gradient launch time against the pixel split (DSRG_WGRAD_KSPLIT is read once per process):
   for k in 0 4 6 8 12; do DSRG_WGRAD_KSPLIT=$k python tools/wgrad_ksplit_probe.py; done"""
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


out = []
for name, H, cin, cout, k, dils in [("conv4_2", 41, 512, 512, 3, [1]), ("conv4_1", 41, 256, 512, 3, [1]), ("fc6x4", 41, 512, 1024, 3, [6, 12, 18, 24]),
                                    ("fc7x4", 41, 1024, 1024, 1, [1] * 4), ("conv3_2", 81, 256, 256, 3, [1]), ("conv3_1", 81, 128, 256, 3, [1])]:
    n = len(dils)
    xs = [torch.randn(16, cin, H, H, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
    gs = [torch.randn(16, cout, H, H, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
    fn = lambda: ops.conv_igemm_wgrad(xs, gs, dils, k)                              # noqa: E731
    for _ in range(3):
        fn()
    out.append("%s %.1f" % (name, np.median([timed(fn) for _ in range(5)])))
print("ksplit %s: " % os.environ.get("DSRG_WGRAD_KSPLIT", "auto") + "  ".join(out), flush=True)
