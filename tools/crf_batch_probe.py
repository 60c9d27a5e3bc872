#!/usr/bin/env python
"""the batched full-resolution CRF on its own (for rocprofv3): `reps` calls of CRF_device_batch over `nb` copies of one 321x321
image (21 labels, log-probability unaries, scale_factor 1, 10 iterations); prints ms per image.  usage: crf_batch_probe.py [nb] [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd import synthetic as S
from dsrg_amd.crf import CRF_device_batch
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H = W = 321
C = 21
ims, uns = [], []
for k in range(nb):
    rng = np.random.default_rng(3000 + k)
    img = S.make_images(rng, 1, size=H)[0] + S.MEAN_PIXEL[:, None, None]
    ims.append(torch.from_numpy(np.ascontiguousarray(np.transpose(img, (1, 2, 0))).astype(np.uint8)))
    lg = S.make_logits(rng, 1, C, H, W, gain=12.0, sigma=12.0)[0]
    e = np.exp(lg - lg.max(0, keepdims=True))
    uns.append(torch.from_numpy(np.log(np.maximum(e / e.sum(0, keepdims=True), 1e-5)).transpose(1, 2, 0).astype(np.float32).copy()))
ims, uns = torch.stack(ims).cuda(), torch.stack(uns).cuda()
for _ in range(3):
    CRF_device_batch(ims, uns, scale_factor=1.0, want="map")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = CRF_device_batch(ims, uns, scale_factor=1.0, want="map")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("batch %d: %.3f ms per image (%.1f images/s), %d distinct images" % (nb, dt / reps / nb * 1e3, nb * reps / dt, nb))
sys.stdout.flush()
