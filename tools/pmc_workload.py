#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (profiles/): a byte-count calibration launch followed by
supervision steps.  The calibration runs dsrg_softmax_forward on a tensor larger than the 256 MiB
Infinity Cache, so its HBM traffic is known (reads n*4 B, writes n*4 B, 4-byte coalesced accesses —
the same access width the hot-path kernels use)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S  # noqa: E402

B = 16
b = S.make_batch(1000, B)
d = lambda a: torch.from_numpy(a).cuda()
logits, images, labels, cues = d(b["logits"]), d(b["images"]), d(b["labels"]), d(b["cues"])
big = torch.randn(2048, 21, 41, 41, device="cuda")          # 289 MB
for _ in range(3):
    ops.softmax_forward(big)                                  # calibration: softmax_fwd_kernel, grid 13448 blocks
torch.cuda.synchronize()
ctx = ops.get_context(B, 21, 41, 41)
for _ in range(12):
    ops.supervision_step(logits, images, labels, cues, ctx=ctx)
torch.cuda.synchronize()
print("calib_bytes_read", big.numel() * 4, "calib_bytes_written", big.numel() * 4)
