#!/usr/bin/env python
"""the batched full-resolution CRF alone (eight 321 x 321 images per call, 21 labels, 10 iterations, lattices built per call): run under
rocprofv3 --kernel-trace --stats for the per-kernel split of modes.crf_fullres.ms_per_image_batch8"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd.crf import CRF_device_batch
rng = np.random.RandomState(0)
H = W = 321
yy, xx = np.mgrid[0:H, 0:W]
im = np.stack([(yy * 255 // H), (xx * 255 // W), ((yy + xx) * 255 // (H + W))], -1).astype(np.float64)
im = np.clip(im + rng.normal(0, 12, im.shape), 0, 255).astype(np.uint8)
p = rng.dirichlet(np.ones(21), size=(H, W)).astype(np.float32)
un = np.log(np.clip(p, 1e-5, 1))
ims8 = torch.from_numpy(np.stack([im] * 8)).cuda()
uns8 = torch.from_numpy(np.stack([un] * 8)).cuda()
for _ in range(3):
    CRF_device_batch(ims8, uns8, scale_factor=1.0, want="map")
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.perf_counter()
for _ in range(n):
    CRF_device_batch(ims8, uns8, scale_factor=1.0, want="map")
torch.cuda.synchronize()
print("ms per image (batch 8): %.4f" % ((time.perf_counter() - t0) / n / 8 * 1e3))
