#!/usr/bin/env python
"""kernel durations of the merged backward launch for one shape (run under rocprofv3 --kernel-trace --stats):
usage: merged_bwd_trace.py cin cout hw k dil"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops
cl = torch.channels_last
cin, cout, hw, k, d = [int(v) for v in sys.argv[1:6]]
x = torch.randn(10, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
g = torch.randn(10, cout, hw, hw, device="cuda").bfloat16().contiguous(memory_format=cl)
w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=cl)
pd = ops.pack_conv_weight(w, for_dgrad=True)
for _ in range(8):
    ops.conv_igemm_backward_residual(g, pd, x, d, k)
torch.cuda.synchronize()
