#!/usr/bin/env python
"""the classifier heads' backward with fc7's ReLU / Dropout backward in the data gradient's store against the plain backward +
four ops.relu_bwd_bias passes (batch 16, 41x41, 1024 channels, 4 branches):  [DSRG_HEAD_TILES=t] python tools/heads_bwd_probe.py"""
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


xs = [torch.relu(torch.randn(16, 1024, 41, 41, device="cuda")).bfloat16().contiguous(memory_format=CL) for _ in range(4)]
w = torch.randn(4, 21, 1024, device="cuda") * 0.05
g = torch.randn(16, 21, 41, 41, device="cuda")


def separate():
    gxs, gw = ops.heads_backward(xs, w, g)
    return [ops.relu_bwd_bias(a, x, 2.0) for a, x in zip(gxs, xs)], gw


fns = {"fused": lambda: ops.heads_backward(xs, w, g, True, 2.0), "separate": separate, "plain": lambda: ops.heads_backward(xs, w, g)}
for fn in fns.values():
    for _ in range(3):
        fn()
t = {k: [] for k in fns}
for _ in range(5):
    for k, fn in fns.items():
        t[k].append(timed(fn))
print("tiles/workgroup %s: " % os.environ.get("DSRG_HEAD_TILES", "default") + "  ".join("%s %.1f us" % (k, np.median(v)) for k, v in t.items()), flush=True)
