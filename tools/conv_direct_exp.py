#!/usr/bin/env python
"""what binds the direct 3x3 kernels (conv_direct.hip): microseconds per launch at batch 16 for the library named by DSRG_LIB
(experiment builds: make -C dsrg_amd/csrc EXP=n EXPSRC=conv_direct exp — 1 no halo fetch, 2 no stores, 4 no MFMAs)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
cl = torch.channels_last
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
out = []
for cin, cout, hw in [(64, 64, 321), (64, 128, 161), (128, 128, 161), (128, 64, 161)]:
    x = torch.randn(16, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).bfloat16().contiguous(memory_format=cl)
    b = torch.randn(cout, device="cuda")
    out.append("%d->%d@%d %.1f" % (cin, cout, hw, t(lambda: ops.conv3x3_direct(x, w, b, True))))
print("%-24s %s" % (os.environ.get("DSRG_LIB", "libdsrg_hip.so"), "   ".join(out)))
