#!/usr/bin/env python
"""the classifier heads' forward (four fc8 branches + sum, float32 weights as three bf16 terms), batch 16, 41 x 41, 1024 channels:
microseconds per launch and a checksum of the output   [DSRG_HEAD_FWD_TILES=1|2] python tools/heads_fwd_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
CL = torch.channels_last
torch.manual_seed(0)
xs = [torch.relu(torch.randn(16, 1024, 41, 41, device="cuda")).bfloat16().contiguous(memory_format=CL) for _ in range(4)]
w = torch.randn(4, 21, 1024, device="cuda") * 0.05
b = torch.randn(4, 21, device="cuda")
for _ in range(3):
    out = ops.heads_forward(xs, w, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.heads_forward(xs, w, b)
e1.record(); torch.cuda.synchronize()
import hashlib
print("tiles per workgroup %s: %.1f us per launch, output sha %s" % (os.environ.get("DSRG_HEAD_FWD_TILES", "2"), e0.elapsed_time(e1) / 20 * 1e3,
      hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]))
