#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes of the full-resolution CRF (bench.py --mode crf-fullres): the byte-count
calibration launch of tools/pmc_workload.py, then the 321x321x21 test-time CRF a few times."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import ops, synthetic as S  # noqa: E402
from dsrg_amd.crf import DenseCRF  # noqa: E402

big = torch.randn(2048, 21, 41, 41, device="cuda")          # 289 MB
for _ in range(3):
    ops.softmax_forward(big)                                  # calibration: softmax_fwd_kernel
torch.cuda.synchronize()
H = W = 321
C = 21
rng = np.random.default_rng(3000 + H)
img = S.make_images(rng, 1, size=H)[0] + S.MEAN_PIXEL[:, None, None]
im = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (1, 2, 0))).astype(np.uint8)).cuda()
logits = S.make_logits(rng, 1, C, H, W, gain=12.0, sigma=12.0)[0]
e = np.exp(logits - logits.max(0, keepdims=True))
un = np.log(np.maximum(e / e.sum(0, keepdims=True), 1e-5)).transpose(1, 2, 0).astype(np.float32)
neg = (-torch.from_numpy(np.ascontiguousarray(un)).cuda()).contiguous()
out = torch.empty((H, W, C), dtype=torch.float32, device="cuda")
crf = DenseCRF(W, H, C)
for _ in range(4):
    crf.set_unary_energy(neg)
    crf.add_pairwise_energy(10, 80.0, 80.0, 13, 13, 13, 3, 3.0, 3.0, im)
    crf.inference(10, out=out)
torch.cuda.synchronize()
print("calib_bytes_read", big.numel() * 4)
